"""Prints the mat-vec grouping decisions of graph_compute for one small decode (GGML_MI355X_DEBUG_GROUP=1)."""
import os; os.environ.setdefault("EH_LAB_PLUGIN", "1")      # lab knobs / stamp kernels live in the --lab build of the plugin only
import sys, os
os.environ["GGML_MI355X_DEBUG_GROUP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
ea = load_package()
gpu = ea.Backend.mi355x(0)
tgt = ea.Model(gpu, "tiny", "q4_k_m", n_ctx=256, seed=1)
toks = list(range(5, 11))
tgt.decode(toks, list(range(len(toks))), want_hidden=True)
