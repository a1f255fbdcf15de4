import os; os.environ.setdefault("EH_LAB_PLUGIN", "1")      # lab knobs / stamp kernels live in the --lab build of the plugin only
import sys, os; sys.path.insert(0,'/root/repo/tests')
os.environ["GGML_MI355X_TRACE_OPS"]="1"
from conftest import load_package
import numpy as np
ea=load_package(); be=ea.Backend.mi355x(0)
t=ea.Model(be,"tiny","q4_k_m",n_ctx=256,seed=5); d=ea.Model(be,"tiny","q4_k_m",n_ctx=256,eagle_of=t,seed=5,accept_p=0.8)
prompt=[int(x) for x in np.random.default_rng(3).integers(5,512,16)]
s=ea.SpecSession(t,d,prompt)
print("=====ROUND", file=sys.stderr, flush=True)
s.rounds(1,n_draft=3)
