// model.cpp -- see model.h.  Synthetic weights + llama / eagle graph construction + decode.
#include "model.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <cstdlib>

namespace eh {

static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------ fp16 helpers (IEEE, round to nearest even)
static inline uint16_t f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    const int e = (int)(ax >> 23) - 127;
    if (e > 15) return (uint16_t)(sign | 0x7c00u);
    const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift = 13; uint32_t base = 0;
    if (e < -14) { shift = 13 + (-14 - e); if (shift > 25) return (uint16_t) sign; } else base = (uint32_t)(e + 14) << 10;
    uint32_t q = m >> shift; const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (base + q));
}
static inline float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) { if (!man) bits = sign; else { int e = -1; uint32_t m = man; do { e++; m <<= 1; } while (!(m & 0x400u)); bits = sign | ((uint32_t)(112 - e) << 23) | ((m & 0x3ffu) << 13); } }
    else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

size_t row_bytes(int type, int64_t k) { return mh::row_size(type, k); }

static bool use_more_bits(int il, int n) { return il < n/8 || il >= 7*n/8 || (il - n/8) % 3 == 2; }   // R/src/llama-quant.cpp:129-131
int weight_type_for(const ModelConfig & c, const char * which, int il) {
    const std::string w = which;
    if (c.ftype == FTYPE_Q8_0) return GGML_TYPE_Q8_0;
    if (w == "output") return GGML_TYPE_Q6_K;                       // llama-quant.cpp:151-166
    if (c.ftype == FTYPE_Q4_0) return GGML_TYPE_Q4_0;
    if ((w == "wv" || w == "down") && use_more_bits(il, c.n_layer)) return GGML_TYPE_Q6_K;   // :235-236, :291-296
    return GGML_TYPE_Q4_K;
}

// ------------------------------------------------------------------ KV cells
bool KVCache::find_slot(const Batch & b) {
    const uint32_t n_tokens = (uint32_t) b.n_tokens();
    if (n_tokens > size) return false;
    uint32_t n_tested = 0;
    while (true) {
        if (head + n_tokens > size) { n_tested += size - head; head = 0; continue; }
        bool found = true;
        for (uint32_t i = 0; i < n_tokens; i++) if (cells[head + i].pos >= 0) { found = false; head += i + 1; n_tested += i + 1; break; }
        if (found) break;
        if (n_tested >= size) return false;
    }
    for (uint32_t k = 0; k < n_tokens; ++k) { cells[head + k].pos = b.pos[k]; cells[head + k].seqs |= b.seq_mask[k]; }
    used += n_tokens;
    return true;
}
uint32_t KVCache::cell_max() const { for (uint32_t i = size; i > 0; --i) if (cells[i-1].pos >= 0 && cells[i-1].seqs) return i; return 0; }
void KVCache::seq_rm(int seq, int32_t p0, int32_t p1) {
    uint32_t new_head = size;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT32_MAX;
    for (uint32_t i = 0; i < size; ++i) if (cells[i].pos >= p0 && cells[i].pos < p1) {
        if (seq < 0) cells[i].seqs = 0;
        else if (cells[i].seqs & (1ull << seq)) cells[i].seqs &= ~(1ull << seq);
        else continue;
        if (!cells[i].seqs) { if (cells[i].pos >= 0) used--; cells[i].pos = -1; if (new_head == size) new_head = i; }
    }
    if (new_head != size && new_head < head) head = new_head;
}
void KVCache::seq_cp(int src, int dst, int32_t p0, int32_t p1) {
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT32_MAX;
    head = 0;
    for (uint32_t i = 0; i < size; ++i) if ((cells[i].seqs & (1ull << src)) && cells[i].pos >= p0 && cells[i].pos < p1) cells[i].seqs |= 1ull << dst;
}
void KVCache::seq_keep(int seq) {
    uint32_t new_head = size;
    for (uint32_t i = 0; i < size; ++i) {
        if (!(cells[i].seqs & (1ull << seq))) { if (cells[i].pos >= 0) used--; cells[i].pos = -1; cells[i].seqs = 0; if (new_head == size) new_head = i; }
        else cells[i].seqs = 1ull << seq;
    }
    if (new_head != size && new_head < head) head = new_head;
}

// ------------------------------------------------------------------ synthetic weights
struct Rng { uint64_t s; explicit Rng(uint64_t seed) : s(seed) {} uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
             double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); } };

// random but valid blocks; `scale` ~ std of the dequantised weights
static void fill_blocks(int type, uint8_t * dst, int64_t rows, int64_t k, uint64_t seed, float scale) {
    const auto tr = mh::traits(type);
    const int64_t nb = rows * (k / tr.blck);
    #pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nb; ++b) {
        Rng r(seed * 0x100000001B3ull + (uint64_t) b);
        uint8_t * p = dst + b * tr.size;
        for (int i = 0; i < tr.size; i += 8) { uint64_t v = r.next(); memcpy(p + i, &v, std::min(8, tr.size - i)); }
        const float u = 0.5f + (float) r.uni();
        uint16_t h;
        // zero-mean weights: K-quants use min = 7.5 (15.5) * scale per sub-block, i.e. x = d*sc*(q - 7.5);
        // Q4_0 (x = d*(q-8), mean -0.5 d) gets a random sign on d.
        auto k4_scales = [&](uint8_t * sc12) {            // encode sc_j == m_j (6 bits each) in the packed 12-byte field
            uint8_t s6[8]; for (int j = 0; j < 8; ++j) s6[j] = (uint8_t)(r.next() & 63);
            for (int j = 0; j < 4; ++j) {
                sc12[j]     = (uint8_t)((s6[j] & 63) | ((s6[j + 4] >> 4) << 6));
                sc12[j + 4] = (uint8_t)((s6[j] & 63) | ((s6[j + 4] >> 4) << 6));
                sc12[j + 8] = (uint8_t)((s6[j + 4] & 0xF) | ((s6[j + 4] & 0xF) << 4));
            }
        };
        switch (type) {
            case GGML_TYPE_Q4_0: h = f2h(scale / 4.6f * u * ((r.next() & 1) ? 1.f : -1.f)); memcpy(p, &h, 2); break;   // q-8 uniform: std 4.6
            case GGML_TYPE_Q8_0: h = f2h(scale / 74.f * u); memcpy(p, &h, 2); break;                                      // int8 uniform: std 74
            case GGML_TYPE_Q4_K: { const float d = scale / (18.f*4.6f) * u; h = f2h(d); memcpy(p, &h, 2); h = f2h(7.5f * h2f(h)); memcpy(p + 2, &h, 2); k4_scales(p + 4); } break;
            case GGML_TYPE_Q5_K: { const float d = scale / (18.f*9.2f) * u; h = f2h(d); memcpy(p, &h, 2); h = f2h(15.5f * h2f(h)); memcpy(p + 2, &h, 2); k4_scales(p + 4); } break;
            case GGML_TYPE_Q6_K: for (int i = 192; i < 208; ++i) p[i] = (uint8_t)(int8_t)((int)(p[i] & 63) - 32);   // scales in [-32,31]
                                 h = f2h(scale / (18.5f*18.5f) * u); memcpy(p + 208, &h, 2); break;
            default: break;
        }
    }
}
// dequantise one row of the LM head (Q6_K or Q8_0 or Q4_0): layouts of R/ggml/src/ggml-common.h:160-320
static void dequant_row(int type, const uint8_t * row, float * y, int64_t k) {
    if (type == GGML_TYPE_Q6_K) {
        for (int64_t i = 0; i < k/256; ++i) {
            const uint8_t * b = row + i*210, * ql = b, * qh = b + 128; const int8_t * sc = (const int8_t *)(b + 192);
            uint16_t dh; memcpy(&dh, b + 208, 2); const float d = h2f(dh);
            float * yy = y + i*256;
            for (int n = 0; n < 2; ++n) {
                for (int l = 0; l < 32; ++l) {
                    const int is = l/16;
                    const int q1 = (int)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32, q2 = (int)((ql[l+32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                    const int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32, q4 = (int)((ql[l+32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                    yy[l] = d*sc[is]*q1; yy[l+32] = d*sc[is+2]*q2; yy[l+64] = d*sc[is+4]*q3; yy[l+96] = d*sc[is+6]*q4;
                }
                yy += 128; ql += 64; qh += 32; sc += 8;
            }
        }
    } else if (type == GGML_TYPE_Q8_0) {
        for (int64_t i = 0; i < k/32; ++i) { const uint8_t * b = row + i*34; uint16_t dh; memcpy(&dh, b, 2); const float d = h2f(dh); for (int j = 0; j < 32; ++j) y[i*32+j] = d*(int8_t) b[2+j]; }
    } else if (type == GGML_TYPE_Q4_0) {
        for (int64_t i = 0; i < k/32; ++i) { const uint8_t * b = row + i*18; uint16_t dh; memcpy(&dh, b, 2); const float d = h2f(dh);
            for (int j = 0; j < 16; ++j) { y[i*32+j] = d*((b[2+j] & 0xF) - 8); y[i*32+16+j] = d*((b[2+j] >> 4) - 8); } }
    }
}
// fc = [ I | 0 ] in the draft's quantised type: row r has a single 1 at column r
static void fill_identity(int type, uint8_t * dst, int64_t rows, int64_t k) {
    const auto tr = mh::traits(type);
    const size_t rb = mh::row_size(type, k);
    memset(dst, 0, rows * rb);
    for (int64_t r = 0; r < rows; ++r) {
        uint8_t * b = dst + r*rb + (r / tr.blck) * tr.size; const int e = (int)(r % tr.blck);
        uint16_t h;
        if (type == GGML_TYPE_Q4_K) {                       // x = d*sc*q - dmin*m : sub-block j = e/32, sc=63 (j<4: scales[j]; else packed)
            const int j = e / 32;
            h = f2h(1.0f/(63*15)); memcpy(b, &h, 2);        // dmin = 0
            if (j < 4) b[4 + j] = 63; else { b[4 + j + 4] = 63 & 0xF; b[4 + j - 4] |= (63 >> 4) << 6; }
            const int g = e / 64, l = e % 32; const bool hi = (e % 64) >= 32;
            b[16 + 32*g + l] |= hi ? (15 << 4) : 15;
        } else if (type == GGML_TYPE_Q8_0) { h = f2h(1.0f/127); memcpy(b, &h, 2); b[2 + e] = 127; }
        else if (type == GGML_TYPE_Q4_0) {                  // (q-8)*d with all other quants at 8 (zero)
            h = f2h(1.0f/7); memcpy(b, &h, 2); memset(b + 2, 0x88, 16);
            if (e < 16) b[2 + e] = (b[2 + e] & 0xF0) | 15; else b[2 + e - 16] = (b[2 + e - 16] & 0x0F) | (15 << 4);
        }
    }
    if (type == GGML_TYPE_Q4_0) {                           // zero blocks must also decode to 0: q = 8 everywhere, d = 0
        for (int64_t r = 0; r < rows; ++r) for (int64_t bi = 0; bi < k/32; ++bi) { uint8_t * b = dst + r*rb + bi*18; uint16_t dh; memcpy(&dh, b, 2); if (!dh) memset(b + 2, 0x88, 16); }
    }
}

Model::~Model() { if (stage_in && be) { be->synchronize(); be->host_free(stage_in); } }

Model * Model::create_synthetic(mh::Backend * be, const ModelConfig & cfg, const SynthOptions & opt, const Model * target) {
    Model * m = new Model;
    m->cfg = cfg; m->be = be; m->logits.be = be; m->hidden.be = be; m->ids_stage.be = be;
    m->wctx.reset(new mh::Ctx(be)); m->wctx->usage = GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
    m->gctx.reset(new mh::Ctx(be)); m->gctx->usage = GGML_BACKEND_BUFFER_USAGE_COMPUTE;
    mh::Ctx & w = *m->wctx;
    // tensor-parallel shard (Megatron pairing on quantised blocks): heads split evenly, n_ff split in whole 256-element
    // super-blocks so that the k-split of ffn_down stays block-aligned
    const int N = cfg.tp_size, R = cfg.tp_rank;
    if (cfg.n_head % N || cfg.n_head_kv % N) { delete m; return nullptr; }
    m->n_head_local = cfg.n_head / N; m->n_head_kv_local = cfg.n_head_kv / N;
    { const int nsb = cfg.n_ff / 256, base = nsb / N, rem = nsb % N; m->n_ff_local = (base + (R < rem ? 1 : 0)) * 256; if (cfg.n_ff % 256) m->n_ff_local = cfg.n_ff / N; }
    const int E = cfg.n_embd, KV = m->n_head_kv_local * cfg.head_dim, Q = m->n_head_local * cfg.head_dim, F = m->n_ff_local, V = cfg.n_vocab;
    // kind 0 random blocks, 1 ones(f32), 2 zeros(f32), 3 identity.  split: 0 whole, 1 rows [lo,hi) of a [gk x grows] tensor,
    // 2 columns (k) [lo,hi) of every row -- a tensor-parallel shard is a slice of the SAME global tensor on every rank
    struct Pending { ggml_tensor * t; int kind; uint64_t seed; float scale; int split; int64_t gk, grows, lo, hi; };
    std::vector<Pending> pend;
    uint64_t sd = opt.seed * 1000003ull + (cfg.eagle ? 7777 : 0);
    const float tiny = opt.predictable ? (getenv("EH_TINY") ? (float) atof(getenv("EH_TINY")) : 1e-4f) : 0.1f;     // residual branches: ~nothing in the predictable model, a few % of the stream otherwise (as in a trained net; O(1) branches make the quantised model chaotic)
    auto mat = [&](const char * which, int il, int64_t gk, int64_t grows, float scale, const char * name, int split = 0, int64_t lo = 0, int64_t hi = 0) {
        const int64_t k = split == 2 ? hi - lo : gk, rows = split == 1 ? hi - lo : grows;
        ggml_tensor * t = w.new_tensor(weight_type_for(cfg, which, il), k, rows, 1, 1, name);
        pend.push_back({t, 0, ++sd, scale, split, gk, grows, lo, hi}); m->weight_bytes += mh::nbytes(t); return t;
    };
    const int Eg = cfg.n_embd, Qg = cfg.n_head * cfg.head_dim, KVg = cfg.n_head_kv * cfg.head_dim, Fg = cfg.n_ff;
    int64_t f0 = 0; { const int nsb = cfg.n_ff / 256, base = nsb / N, rem = nsb % N; for (int r = 0; r < R; ++r) f0 += (int64_t)(base + (r < rem ? 1 : 0)) * 256; if (cfg.n_ff % 256) f0 = (int64_t) R * (cfg.n_ff / N); }
    const int64_t q0 = (int64_t) R * Q, kv0 = (int64_t) R * KV;
    const int sp_r = N > 1 ? 1 : 0, sp_k = N > 1 ? 2 : 0;
    char nm[64];
    m->layers.resize(cfg.n_layer);
    if (cfg.eagle) {
        m->fc = w.new_tensor(weight_type_for(cfg, "fc", 0), 2*E, E, 1, 1, "fc.weight"); m->weight_bytes += mh::nbytes(m->fc);
        pend.push_back({m->fc, opt.predictable ? 3 : 0, ++sd, 0.02f, 0, 0, 0, 0, 0});
        m->fc_b = w.new_tensor(GGML_TYPE_F32, E, 1, 1, 1, "fc.bias"); pend.push_back({m->fc_b, 2, 0, 0, 0, 0, 0, 0, 0});
    }
    for (int il = 0; il < cfg.n_layer; ++il) {
        Layer & L = m->layers[il];
        if (!cfg.eagle) { snprintf(nm, sizeof nm, "blk.%d.attn_norm.weight", il); L.attn_norm = w.new_tensor(GGML_TYPE_F32, E, 1, 1, 1, nm); pend.push_back({L.attn_norm, 1, 0, 0, 0, 0, 0, 0, 0}); }
        snprintf(nm, sizeof nm, "blk.%d.attn_q.weight", il);      L.wq = mat("wq", il, Eg, Qg, 0.02f, nm, sp_r, q0, q0 + Q);
        snprintf(nm, sizeof nm, "blk.%d.attn_k.weight", il);      L.wk = mat("wk", il, Eg, KVg, 0.02f, nm, sp_r, kv0, kv0 + KV);
        snprintf(nm, sizeof nm, "blk.%d.attn_v.weight", il);      L.wv = mat("wv", il, Eg, KVg, 0.02f, nm, sp_r, kv0, kv0 + KV);
        snprintf(nm, sizeof nm, "blk.%d.attn_output.weight", il); L.wo = mat("wo", il, Qg, Eg, 0.02f * tiny, nm, sp_k, q0, q0 + Q);
        snprintf(nm, sizeof nm, "blk.%d.ffn_norm.weight", il);    L.ffn_norm = w.new_tensor(GGML_TYPE_F32, E, 1, 1, 1, nm); pend.push_back({L.ffn_norm, 1, 0, 0, 0, 0, 0, 0, 0});
        snprintf(nm, sizeof nm, "blk.%d.ffn_gate.weight", il);    L.gate = mat("gate", il, Eg, Fg, 0.02f, nm, sp_r, f0, f0 + F);
        snprintf(nm, sizeof nm, "blk.%d.ffn_up.weight", il);      L.up = mat("up", il, Eg, Fg, 0.02f, nm, sp_r, f0, f0 + F);
        snprintf(nm, sizeof nm, "blk.%d.ffn_down.weight", il);    L.down = mat("down", il, Fg, Eg, 0.02f * tiny, nm, sp_k, f0, f0 + F);
    }
    if (!cfg.eagle) {
        m->output_norm = w.new_tensor(GGML_TYPE_F32, E, 1, 1, 1, "output_norm.weight"); pend.push_back({m->output_norm, 1, 0, 0, 0, 0, 0, 0, 0});
        m->output = mat("output", 0, E, V, 0.02f, "output.weight");
    } else {
        m->lm_head_from = target;
    }
    // KV cache: f16, one tensor per layer, cleared like llama_kv_cache_init does (R/src/llama-kv-cache.cpp:27-118)
    m->kv.init(cfg.n_ctx);
    for (int il = 0; il < cfg.n_layer; ++il) {
        snprintf(nm, sizeof nm, "cache_k_l%d", il); m->k_l.push_back(w.new_tensor(GGML_TYPE_F16, (int64_t) KV * cfg.n_ctx, 1, 1, 1, nm));
        snprintf(nm, sizeof nm, "cache_v_l%d", il); m->v_l.push_back(w.new_tensor(GGML_TYPE_F16, (int64_t) KV * cfg.n_ctx, 1, 1, 1, nm));
    }
    if (!w.alloc()) { delete m; return nullptr; }
    for (auto b : w.buffers) b->iface.clear(b, 0);

    // fill + upload, one tensor at a time through set_tensor (what llama_model_loader::load_all_data does)
    std::vector<uint8_t> stage, gstage;
    std::vector<uint8_t> out_rows;                          // kept for the predictable embedding construction
    for (auto & p : pend) {
        const size_t n = mh::nbytes(p.t);
        stage.resize(n);
        if (p.kind == 0 && p.split == 0) fill_blocks(p.t->type, stage.data(), p.t->ne[1], p.t->ne[0], p.seed, p.scale);
        else if (p.kind == 0) {                               // generate the global tensor, keep this rank's slice
            const auto tr = mh::traits(p.t->type);
            const size_t grb = mh::row_size(p.t->type, p.gk);
            gstage.resize((size_t) p.grows * grb);
            fill_blocks(p.t->type, gstage.data(), p.grows, p.gk, p.seed, p.scale);
            if (p.split == 1) memcpy(stage.data(), gstage.data() + (size_t) p.lo * grb, n);
            else {
                const size_t lrb = mh::row_size(p.t->type, p.hi - p.lo), off = (size_t)(p.lo / tr.blck) * tr.size;
                for (int64_t r = 0; r < p.grows; ++r) memcpy(stage.data() + (size_t) r * lrb, gstage.data() + (size_t) r * grb + off, lrb);
            }
        }
        else if (p.kind == 1) { float * f = (float *) stage.data(); for (size_t i = 0; i < n/4; ++i) f[i] = 1.0f; }
        else if (p.kind == 2) memset(stage.data(), 0, n);
        else fill_identity(p.t->type, stage.data(), p.t->ne[1], p.t->ne[0]);
        w.set(p.t, stage.data(), 0, n);
        if (p.t == m->output) out_rows = stage;
    }
    // token embeddings (host side)
    m->tok_embd.resize((size_t) V * E);
    Rng r(opt.seed ^ 0xABCDEF1234ull ^ (cfg.eagle ? 99 : 0));
    if (!opt.predictable) {
        for (size_t i = 0; i < m->tok_embd.size(); ++i) { const double u1 = r.uni() + 1e-12, u2 = r.uni(); m->tok_embd[i] = f2h((float)(std::sqrt(-2*std::log(u1)) * std::cos(6.283185307179586*u2))); }
    } else {
        // next(t) = perm(t):  e_t is (a multiple of) row perm(t) of the LM head, so argmax_v <W_v, norm(e_t)> = perm(t).
        // The draft sees relu() after fc, so it gets the positive part of that row -- or, with probability 1-accept_p,
        // of a wrong row, which makes it mispredict exactly those tokens.
        const Model * src = cfg.eagle ? target : m;
        std::vector<uint8_t> tgt_rows;
        const uint8_t * rows = out_rows.data();
        const int otype = src->output->type;
        const size_t rb = mh::row_size(otype, E);
        if (cfg.eagle) { tgt_rows.resize(mh::nbytes(src->output)); src->wctx->get(src->output, tgt_rows.data(), 0, tgt_rows.size()); rows = tgt_rows.data(); }
        std::vector<int32_t> perm(V); std::iota(perm.begin(), perm.end(), 0);
        Rng pr(opt.seed * 31 + 5);                         // same permutation for target and draft
        for (int i = V - 1; i > 0; --i) { const int j = (int)(pr.next() % (uint64_t)(i + 1)); std::swap(perm[i], perm[j]); }
        Rng br(opt.seed * 77 + 1);
        std::vector<uint8_t> bad(V, 0); std::vector<int32_t> wrong(V, 0);
        for (int t = 0; t < V; ++t) { bad[t] = cfg.eagle && br.uni() >= opt.accept_p; wrong[t] = (int32_t)(br.next() % (uint64_t) V); }
        #pragma omp parallel
        {
            std::vector<float> wrow(E);
            #pragma omp for schedule(static)
            for (int t = 0; t < V; ++t) {
                const int srcrow = bad[t] ? perm[wrong[t]] : perm[t];
                dequant_row(otype, rows + (size_t) srcrow * rb, wrow.data(), E);
                double ss = 0; for (int i = 0; i < E; ++i) ss += (double) wrow[i]*wrow[i];
                const float a = (float)(1.0 / std::sqrt(ss / E + 1e-30));
                for (int i = 0; i < E; ++i) { float v = wrow[i] * a; if (cfg.eagle && v < 0) v = 0; m->tok_embd[(size_t) t*E + i] = f2h(v); }
            }
        }
    }
    return m;
}

// fp16 -> fp32 row conversion of the host-resident token embeddings (llama_set_inputs does the same through ggml_get_rows on the CPU
// backend): F16C when the host has it (exact, so identical to the scalar routine), scalar otherwise
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("f16c,avx"))) static void h2f_row_f16c(const uint16_t * src, float * dst, int n) {
    int i = 0;
    for (; i + 8 <= n; i += 8) _mm256_storeu_ps(dst + i, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(src + i))));
    for (; i < n; ++i) dst[i] = h2f(src[i]);
}
#endif
static void h2f_row(const uint16_t * src, float * dst, int n) {
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("f16c") && __builtin_cpu_supports("avx");
    if (fast) { h2f_row_f16c(src, dst, n); return; }
#endif
    for (int i = 0; i < n; ++i) dst[i] = h2f(src[i]);
}

// ------------------------------------------------------------------ graph of one forward pass (build_llama :1647 / build_eagle :1839)
void Model::tp_node_hook(void * user, const ggml_tensor * t, void *) {
    Model * m = (Model *) user;
    m->allreduce(m->allreduce_user, t->data, mh::nelements(t)); m->n_allreduce++;
}
void Model::build_forward(mh::Ctx & g, const StepIO & io, bool tp, std::vector<Cut> * cuts,
                          ggml_tensor *& result_norm, ggml_tensor *& result_output, ggml_tensor *& result_argmax) {
    const int E = cfg.n_embd, H = n_head_local, Hkv = n_head_kv_local, D = cfg.head_dim, n_ctx = cfg.n_ctx;
    const int KVd = Hkv * D;
    (void) E;
    ggml_tensor * inpL = io.embd, * cur;
    if (cfg.eagle) {                                           // build_eagle :1863-1870
        ggml_tensor * embd_hs = g.concat(io.embd, io.hidd, 0);
        cur = g.mul_mat(fc, embd_hs);
        if (fc_b) cur = g.add(cur, fc_b);
        inpL = g.unary(cur, GGML_UNARY_OP_RELU);
    }
    const float kq_scale = 1.0f / sqrtf((float) D);
    char nm[64];
    int n_tok = io.T;
    for (int il = 0; il < cfg.n_layer; ++il) {
        const Layer & L = layers[il];
        if (il > 0 && il < (int) forced_layer_inp.size() && forced_layer_inp[il]) {
            // parity tooling (tests/test_teacher_forced_gpu.py): this layer reads a host-provided input (another backend's l_out of the
            // layer below) instead of the graph's own, so that differences cannot compound across layers
            ggml_tensor * f = g.new_tensor(GGML_TYPE_F32, cfg.n_embd, n_tok, 1, 1, "forced_inp");
            f->flags |= GGML_TENSOR_FLAG_INPUT;
            forced_tensors.push_back({ f, forced_layer_inp[il] });
            inpL = f;
        }
        ggml_tensor * inpSA = inpL;
        cur = g.rms_norm(inpL, cfg.rms_eps);
        if (L.attn_norm) cur = g.mul(cur, L.attn_norm);
        snprintf(nm, sizeof nm, "attn_norm-%d", il); g.set_name(cur, nm);
        ggml_tensor * Qcur = g.mul_mat(L.wq, cur);
        Qcur = g.rope_ext(g.reshape(Qcur, D, H, n_tok), io.pos, nullptr, D, 0, 0, cfg.rope_base, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f);
        snprintf(nm, sizeof nm, "Qcur-%d", il); g.set_name(Qcur, nm);
        ggml_tensor * Kcur = g.mul_mat(L.wk, cur);
        Kcur = g.rope_ext(g.reshape(Kcur, D, Hkv, n_tok), io.pos, nullptr, D, 0, 0, cfg.rope_base, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f);
        snprintf(nm, sizeof nm, "Kcur-%d", il); g.set_name(Kcur, nm);
        ggml_tensor * Vcur = g.mul_mat(L.wv, cur);
        snprintf(nm, sizeof nm, "Vcur-%d", il); g.set_name(Vcur, nm);
        // llm_build_kv_store :228-270
        ggml_tensor * k_view = g.view_1d(k_l[il], (int64_t) n_tok * KVd, (size_t) KVd * 2 * io.kv_head);
        g.cpy(Kcur, k_view);
        ggml_tensor * v_view = g.view_2d(v_l[il], n_tok, KVd, (size_t) n_ctx * 2, (size_t) io.kv_head * 2);
        g.cpy(g.transpose(Vcur), v_view);
        // llm_build_kqv :706-828 (no flash attention)
        ggml_tensor * q = g.permute(Qcur, 0, 2, 1, 3);
        ggml_tensor * k = g.view_3d(k_l[il], D, io.n_kv, Hkv, (size_t) KVd * 2, (size_t) D * 2, 0);
        ggml_tensor * kq = g.mul_mat(k, q);
        kq->op_params[0] = GGML_PREC_F32;
        kq = g.soft_max_ext(kq, io.mask, kq_scale, 0.0f);
        ggml_tensor * v = g.view_3d(v_l[il], io.n_kv, D, Hkv, (size_t) n_ctx * 2, (size_t) n_ctx * D * 2, 0);
        ggml_tensor * kqv = g.mul_mat(v, kq);
        cur = g.cont_2d(g.permute(kqv, 0, 2, 1, 3), (int64_t) D * H, n_tok);
        cur = g.mul_mat(L.wo, cur);                            // TP: partial sum over this rank's heads
        if (tp && cuts) { cur->flags |= GGML_TENSOR_FLAG_OUTPUT; cuts->push_back({ (int) g.nodes.size(), cur }); }     // (OUTPUT: no fusion may swallow a tensor that is all-reduced in place)
        if (il == cfg.n_layer - 1) {                           // skip unused tokens :1737-1743
            n_tok = io.n_outputs;
            cur = g.get_rows(cur, io.out_ids);
            inpSA = g.get_rows(inpSA, io.out_ids);
        }
        ggml_tensor * ffn_inp = g.add(cur, inpSA);
        cur = g.rms_norm(ffn_inp, cfg.rms_eps);
        cur = g.mul(cur, L.ffn_norm);
        ggml_tensor * gate = g.mul_mat(L.gate, cur);             // llm_build_ffn, LLM_FFN_SILU / LLM_FFN_PAR, in DFS order
        gate = g.unary(gate, GGML_UNARY_OP_SILU);
        ggml_tensor * up = g.mul_mat(L.up, cur);
        cur = g.mul(gate, up);
        cur = g.mul_mat(L.down, cur);                          // TP: partial sum over this rank's slice of n_ff
        if (tp && cuts) { cur->flags |= GGML_TENSOR_FLAG_OUTPUT; cuts->push_back({ (int) g.nodes.size(), cur }); }
        cur = g.add(cur, ffn_inp);
        snprintf(nm, sizeof nm, "l_out-%d", il); g.set_name(cur, nm);
        inpL = cur;
    }
    result_norm = nullptr; result_output = nullptr; result_argmax = nullptr;
    const bool head_here = !tp || cfg.tp_rank == 0;          // TP: the LM head (and the hidden-state channel) live on rank 0
    if (head_here) {
        cur = g.rms_norm(inpL, cfg.rms_eps);
        if (output_norm) cur = g.mul(cur, output_norm);
        g.set_name(cur, "result_norm");
        result_norm = cur;
        ggml_tensor * head = cfg.eagle ? lm_head_from->output : output;
        cur = g.mul_mat(head, cur);
        g.set_name(cur, "result_output");
        result_output = cur;
        result_norm->flags |= GGML_TENSOR_FLAG_OUTPUT; result_output->flags |= GGML_TENSOR_FLAG_OUTPUT;
        if (!want_logits) { result_argmax = g.argmax(result_output); g.set_name(result_argmax, "result_argmax"); result_argmax->flags |= GGML_TENSOR_FLAG_OUTPUT; }
    }
}

// ------------------------------------------------------------------ decode
// llama_decode in two halves so that a driver can overlap host work with the GPU: decode_prepare() does everything that depends only
// on the batch's SHAPE (positions, seq ids, output flags): KV slots, graph build, allocation, the pos / mask / out_ids part of the input
// image; decode() then only gathers the embeddings (+ hidden rows), uploads, launches and waits.  The speculative driver prepares the
// verification batch while the draft chain is still running on the device (its token ids are the only thing it has to wait for).
static bool same_shape(const Batch & a, const Batch & b) {
    return a.pos == b.pos && a.seq_first == b.seq_first && a.seq_mask == b.seq_mask && a.logits == b.logits;
}
void Model::decode_abandon() { if (pend.valid) { kv = pend.kv_saved; pend.valid = false; } }
int Model::decode_prepare(const Batch & b) {
    decode_abandon();
    const double t0 = now_us();
    Pending & P = pend;
    const int T = b.n_tokens();
    if (T <= 0) return -1;
    // cells claimed by find_slot are given back when the decode fails (llama_kv_slot_restorer, R/src/llama.cpp:9518-9546)
    P.kv_saved = kv;
    if (!kv.find_slot(b)) return 1;
    const uint32_t pad = 32;                                 // llama_kv_cache_get_padding without flash-attn
    kv.n = std::min(kv.size, std::max(pad, (kv.cell_max() + pad - 1) / pad * pad));
    const int n_kv = (int) kv.n, kv_head = (int) kv.head;
    static const bool force_tp = getenv("EH_FORCE_TP") != nullptr;     // run the segmented path + collectives even with one rank (single-GPU rehearsal)
    P.tp = cfg.tp_size > 1 || (force_tp && allreduce && !cfg.eagle);
    if (P.tp && !allreduce) { kv = P.kv_saved; return -5; }
    const int E = cfg.n_embd;
    P.cuts.clear();
    n_outputs = 0; out_ids.clear();
    for (int i = 0; i < T; ++i) if (b.logits[i]) { out_ids.push_back(i); n_outputs++; }
    if (n_outputs == 0) { out_ids.push_back(T - 1); n_outputs = 1; }

    mh::Ctx & g = *gctx;
    g.reset_graph();
    P.dev_tokens = dev_ids && dev_table && !cfg.eagle && !P.tp && dev_ids->ne[0] == T;
    P.inp_embd = P.dev_tokens ? nullptr : g.new_tensor(GGML_TYPE_F32, E, T, 1, 1, "inp_embd");
    P.inp_hidd = cfg.eagle ? g.new_tensor(GGML_TYPE_F32, E, T, 1, 1, "inp_hidd") : nullptr;
    P.inp_pos  = g.new_tensor(GGML_TYPE_I32, T, 1, 1, 1, "inp_pos");
    const int Tpad = (T + GGML_KQ_MASK_PAD - 1) / GGML_KQ_MASK_PAD * GGML_KQ_MASK_PAD;
    P.kq_mask  = g.new_tensor(GGML_TYPE_F32, n_kv, Tpad, 1, 1, "KQ_mask");
    P.inp_out  = g.new_tensor(GGML_TYPE_I32, n_outputs, 1, 1, 1, "inp_out_ids");
    for (ggml_tensor * t : {P.inp_embd, P.inp_hidd, P.inp_pos, P.kq_mask, P.inp_out}) if (t) t->flags |= GGML_TENSOR_FLAG_INPUT;

    ggml_tensor * embd = P.inp_embd;
    if (P.dev_tokens) embd = g.get_rows((ggml_tensor *) dev_table, (ggml_tensor *) dev_ids);      // llm_build_inp_embd's GET_ROWS branch (R/src/llama.cpp:495-503), indices already on the device
    StepIO io{ embd, P.inp_hidd, P.inp_pos, P.kq_mask, P.inp_out, T, n_outputs, n_kv, kv_head };
    P.result_norm = nullptr; P.result_output = nullptr; P.result_argmax = nullptr;
    forced_tensors.clear();
    build_forward(g, io, P.tp, &P.cuts, P.result_norm, P.result_output, P.result_argmax);
    P.head_here = !P.tp || cfg.tp_rank == 0;                 // TP: the LM head (and the hidden-state channel) live on rank 0
    last_n_nodes = (int) g.nodes.size();
    if (!g.alloc()) { kv = P.kv_saved; return -3; }
    const double t1 = now_us();

    // ---- inputs (llama_set_inputs, R/src/llama-context.cpp:61-210).  The input tensors were created first, so they sit
    // back to back in the compute buffer: their host image is assembled in page-locked memory and goes up as ONE
    // asynchronous copy ordered before the graph (the reference issues one blocking ggml_backend_tensor_set per input).
    ggml_tensor * inputs[5] = { P.inp_embd, P.inp_hidd, P.inp_pos, P.kq_mask, P.inp_out };
    P.span = 0; P.packed = true; P.base = nullptr;
    for (ggml_tensor * t : inputs) if (t && !P.base) P.base = t;
    for (ggml_tensor * t : inputs) if (t) {
        const ptrdiff_t off = (char *) t->data - (char *) P.base->data;
        if (t->buffer != P.base->buffer || off < 0 || (size_t) off > ((size_t) 64 << 20)) { P.packed = false; break; }
        P.span = std::max(P.span, (size_t) off + mh::nbytes(t));
    }
    if (P.packed && P.span > stage_cap) {
        if (stage_in) { be->synchronize(); be->host_free(stage_in); }
        stage_cap = P.span + P.span/2 + 4096; stage_in = (char *) be->host_alloc(stage_cap);
    }
    if (P.packed) {      // shape-dependent part of the image: positions, output rows, mask
        auto host_of = [&](ggml_tensor * t) -> char * { return stage_in + ((char *) t->data - (char *) P.base->data); };
        memcpy(host_of(P.inp_pos), b.pos.data(), (size_t) T * 4);
        memcpy(host_of(P.inp_out), out_ids.data(), (size_t) n_outputs * 4);
        float * mask = (float *) host_of(P.kq_mask);
        std::fill(mask, mask + (size_t) n_kv * T, -INFINITY);        // rows T..Tpad-1 are padding no kernel reads (R pads for flash-attn only)
        for (int j = 0; j < T; ++j) {
            const uint64_t sbit = 1ull << b.seq_first[j]; const int32_t pos = b.pos[j];
            float * row = mask + (size_t) j * n_kv;
            for (int i = 0; i < n_kv; ++i) if ((kv.cells[i].seqs & sbit) && kv.cells[i].pos <= pos) row[i] = 0.0f;
        }
    }
    P.shape = b; P.shape.token.clear(); P.shape.hidd.clear();
    P.T = T; P.n_kv = n_kv; P.want_logits = want_logits;
    P.valid = true;
    t_build_us += t1 - t0; t_upload_us += now_us() - t1;
    return 0;
}

int Model::decode(const Batch & b, bool want_hidden) {
    const int T = b.n_tokens();
    if (T <= 0) return -1;
    if (cfg.eagle && (int) b.hidd.size() != T * cfg.n_embd) { decode_abandon(); return -2; }
    const bool dev = dev_ids && dev_table && !cfg.eagle && cfg.tp_size == 1 && dev_ids->ne[0] == T;      // tokens live on the device: Batch::token is a placeholder
    if (!dev) for (int i = 0; i < T; ++i) if (b.token[i] < 0 || b.token[i] >= cfg.n_vocab) { decode_abandon(); return -6; }       // llama_decode: "invalid token" (R/src/llama.cpp:9500)
    if (!(pend.valid && pend.want_logits == want_logits && pend.dev_tokens == dev && same_shape(pend.shape, b))) {
        const int rc = decode_prepare(b);
        if (rc) return rc;
    }
    Pending & P = pend;
    P.valid = false;
    const double t1 = now_us();
    const int E = cfg.n_embd, n_kv = P.n_kv;
    mh::Ctx & g = *gctx;
    static thread_local std::vector<char> unpacked;
    auto host_of = [&](ggml_tensor * t) -> char * {
        if (P.packed) return stage_in + ((char *) t->data - (char *) P.base->data);
        unpacked.resize(mh::nbytes(t)); return unpacked.data();
    };
    auto flush = [&](ggml_tensor * t, char * h) { if (!P.packed) g.set(t, h, 0, mh::nbytes(t)); };
    if (P.inp_embd) {
        float * embd = (float *) host_of(P.inp_embd);
        for (int i = 0; i < T; ++i) { const uint16_t * src = tok_embd.data() + (size_t) b.token[i] * E; h2f_row(src, embd + (size_t) i * E, E); }
        flush(P.inp_embd, (char *) embd);
    }
    if (P.inp_hidd) { char * h = host_of(P.inp_hidd); memcpy(h, b.hidd.data(), (size_t) T * E * 4); flush(P.inp_hidd, h); }
    if (!P.packed) {
        { char * h = host_of(P.inp_pos); memcpy(h, b.pos.data(), (size_t) T * 4); flush(P.inp_pos, h); }
        { char * h = host_of(P.inp_out); memcpy(h, out_ids.data(), (size_t) n_outputs * 4); flush(P.inp_out, h); }
        float * mask = (float *) host_of(P.kq_mask);
        std::fill(mask, mask + (size_t) n_kv * T, -INFINITY);
        for (int j = 0; j < T; ++j) {
            const uint64_t sbit = 1ull << b.seq_first[j]; const int32_t pos = b.pos[j];
            float * row = mask + (size_t) j * n_kv;
            for (int i = 0; i < n_kv; ++i) if ((kv.cells[i].seqs & sbit) && kv.cells[i].pos <= pos) row[i] = 0.0f;
        }
        flush(P.kq_mask, (char *) mask);
    }
    if (P.packed) g.set_async(P.base, stage_in, 0, P.span);
    for (auto & f : forced_tensors) g.set_async(f.first, f.second, 0, mh::nbytes(f.first));      // (teacher forcing: valid for this decode only)
    forced_layer_inp.clear();
    const double t2 = now_us();

    // ---- compute, then the outputs by asynchronous copies into page-locked memory, one wait for everything
    enum ggml_status st = GGML_STATUS_SUCCESS;
    if (!P.tp) st = g.compute_async();
    else if (tp_in_graph && !P.cuts.empty() && [&] { P.hook_nodes.clear(); for (auto & c : P.cuts) P.hook_nodes.push_back(c.t);
                                                    return be->set_node_hooks(P.hook_nodes.data(), (int) P.hook_nodes.size(), tp_node_hook, this); }()) {
        // the plugin calls back from inside graph_compute after every cut tensor: the all-reduces are enqueued on its stream between
        // the launches, the whole forward is ONE submission (EH_TP_SEGMENTS=1: the segment loop below)
        st = g.compute_async();
        be->set_node_hooks(nullptr, 0, nullptr, nullptr);
    } else {
        int n0 = 0;
        for (auto & c : P.cuts) {
            st = g.compute_range(n0, c.node_end); if (st != GGML_STATUS_SUCCESS) break;
            allreduce(allreduce_user, c.t->data, mh::nelements(c.t)); n_allreduce++;
            n0 = c.node_end;
        }
        if (st == GGML_STATUS_SUCCESS) st = g.compute_range(n0, (int) g.nodes.size());
    }
    const double t3a = now_us();
    if (st == GGML_STATUS_SUCCESS && P.head_here && P.result_argmax) {
        logits.clear();
        if (ids_stage.size() < (size_t) n_outputs) ids_stage.resize((size_t) n_outputs + 64);
        g.get_async(P.result_argmax, ids_stage.data(), 0, (size_t) n_outputs * 4);
        if (want_hidden) { hidden.resize((size_t) n_outputs * E); g.get_async(P.result_norm, hidden.data(), 0, hidden.size() * 4); }
    } else if (st == GGML_STATUS_SUCCESS && P.head_here) {
        topk_k = 0;
        if (want_topk > 0 && n_outputs > 0) {          // the k best of every row, selected on the device: 8 k bytes per row come back instead of n_vocab floats
            topk_ids.resize((size_t) n_outputs * want_topk); topk_vals.resize((size_t) n_outputs * want_topk);
            if (want_hidden) { hidden.resize((size_t) n_outputs * E); g.get_async(P.result_norm, hidden.data(), 0, hidden.size() * 4); }
            if (be->top_k(P.result_output, nullptr, n_outputs, want_topk, topk_ids.data(), topk_vals.data())) { topk_k = want_topk; logits.clear(); }
        }
        if (!topk_k) {
            logits.resize((size_t) n_outputs * cfg.n_vocab);
            g.get_async(P.result_output, logits.data(), 0, logits.size() * 4);
            if (want_hidden) { hidden.resize((size_t) n_outputs * E); g.get_async(P.result_norm, hidden.data(), 0, hidden.size() * 4); }
        }
    } else { logits.clear(); hidden.clear(); }
    be->synchronize();
    const double t3 = now_us();
    g.t_issue_us += t3a - t2; g.t_wait_us += t3 - t3a;
    if (st != GGML_STATUS_SUCCESS) { kv = P.kv_saved; return st == GGML_STATUS_ABORTED ? 2 : -4; }
    argmax_ids.clear();
    last_norm = DevRows(); last_argmax = DevRows();
    if (P.head_here && P.result_argmax) {
        const int32_t * p = (const int32_t *) ids_stage.data(); argmax_ids.assign(p, p + n_outputs);
        if (!P.tp && P.result_norm && n_outputs == T) {        // row r = batch token r
            last_norm = DevRows{ P.result_norm->data, P.result_norm->buffer, n_outputs }; last_argmax = DevRows{ P.result_argmax->data, P.result_argmax->buffer, n_outputs };
        }
    }
    kv.head += T;
    if (kv.head >= kv.size) kv.head = 0;
    const double t4 = now_us();
    t_upload_us += t2 - t1; t_compute_us += t3 - t2; t_download_us += t4 - t3; n_decode++;
    return 0;
}

bool Model::topk_ith(int i, const int32_t ** ids, const float ** vals) const {
    if (topk_k <= 0) return false;
    for (int r = 0; r < n_outputs; ++r) if (out_ids[r] == i) { *ids = topk_ids.data() + (size_t) r * topk_k; *vals = topk_vals.data() + (size_t) r * topk_k; return true; }
    return false;
}
int Model::argmax_ith(int i) const {
    for (int r = 0; r < n_outputs; ++r) if (out_ids[r] == i) {
        if (!argmax_ids.empty()) return argmax_ids[r];
        if (logits.size() == 0) return -1;
        const float * v = logits.data() + (size_t) r * cfg.n_vocab; int b = 0; float m = v[0];
        for (int j = 1; j < cfg.n_vocab; ++j) if (v[j] > m) { m = v[j]; b = j; }
        return b;
    }
    return -1;
}
int Model::fetch_last_hidden(float * dst, int rows) {
    if (!last_norm.data || rows > last_norm.rows) return -1;
    ggml_tensor t; memset(&t, 0, sizeof(t));
    t.type = GGML_TYPE_F32; t.ne[0] = cfg.n_embd; t.ne[1] = rows; t.ne[2] = t.ne[3] = 1;
    t.nb[0] = 4; t.nb[1] = (size_t) cfg.n_embd * 4; t.nb[2] = t.nb[3] = t.nb[1] * rows;
    t.data = last_norm.data; t.buffer = last_norm.buffer;
    be->synchronize();
    t.buffer->iface.get_tensor(t.buffer, &t, dst, 0, (size_t) rows * cfg.n_embd * 4);
    return 0;
}
// device copy of token_embd (f16 [n_embd, n_vocab]), made on first use
const ggml_tensor * Model::tok_embd_device() {
    if (!tok_embd_dev) {
        ectx.reset(new mh::Ctx(be)); ectx->usage = GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
        tok_embd_dev = ectx->new_tensor(GGML_TYPE_F16, cfg.n_embd, cfg.n_vocab, 1, 1, "token_embd.weight");
        if (!ectx->alloc()) { tok_embd_dev = nullptr; ectx.reset(); return nullptr; }
        ectx->set(tok_embd_dev, tok_embd.data(), 0, tok_embd.size() * 2);
    }
    return tok_embd_dev;
}
// ------------------------------------------------------------------ fused greedy draft chain
int Model::decode_chain(const Batch & first, int n_steps, std::vector<int32_t> & ids, bool defer_wait) {
    const double t0 = now_us();
    decode_abandon();
    ids.clear(); chain_ids = nullptr;
    const int T0 = first.n_tokens(), E = cfg.n_embd;
    if (!cfg.eagle || cfg.tp_size > 1 || T0 <= 0 || n_steps < 1) return -1;
    const bool dev_first = first_feat.data && first_ids.data && T0 <= first_feat.rows && T0 <= first_ids.rows;      // step 0 reads the target's rows on the device
    const DevRows dfeat = first_feat, dids = first_ids;
    first_feat = DevRows(); first_ids = DevRows();
    if (!dev_first && (int) first.hidd.size() != T0 * E) return -2;
    for (int i = 0; i < T0; ++i) if (first.token[i] < 0 || first.token[i] >= cfg.n_vocab) return -6;
    if (!tok_embd_device()) return -3;                         // the reference keeps token_embd on the host; the fused loop needs it next to the arg-max
    // ---- KV slots of every step (positions are known in advance: a chain)
    std::vector<Batch> bs(n_steps);
    std::vector<int> heads(n_steps);
    bs[0] = first;
    const int32_t pos_last = first.pos[T0 - 1], seq = first.seq_first[T0 - 1];
    const KVCache kv_saved = kv;
    for (int j = 0; j < n_steps; ++j) {
        if (j > 0) { bs[j].clear(); bs[j].add(0, pos_last + j, seq, true); }
        if (!kv.find_slot(bs[j])) { kv = kv_saved; return 1; }
        heads[j] = (int) kv.head;
        kv.head += bs[j].n_tokens(); if (kv.head >= kv.size) kv.head = 0;
    }
    const uint32_t pad = 32;
    kv.n = std::min(kv.size, std::max(pad, (kv.cell_max() + pad - 1) / pad * pad));
    const int n_kv = (int) kv.n;

    mh::Ctx & g = *gctx;
    g.reset_graph();
    // ---- inputs of all steps first (they go up as one copy), then the graphs
    struct In { ggml_tensor * embd, * hidd, * pos, * mask, * out; int T, Tpad; };
    std::vector<In> in(n_steps);
    std::vector<ggml_tensor *> inputs;
    // [last token of the first batch, arg-max of step 0, 1, ...]: slot 0 goes up with the inputs, every step's ARGMAX writes its slot (below)
    chain_ids = g.new_tensor(GGML_TYPE_I32, 1 + n_steps, 1, 1, 1, "chain_ids");
    chain_ids->flags |= GGML_TENSOR_FLAG_INPUT; inputs.push_back(chain_ids);
    for (int j = 0; j < n_steps; ++j) {
        const int T = bs[j].n_tokens(), Tpad = (T + GGML_KQ_MASK_PAD - 1) / GGML_KQ_MASK_PAD * GGML_KQ_MASK_PAD;
        in[j].T = T; in[j].Tpad = Tpad;
        in[j].embd = j == 0 && !dev_first ? g.new_tensor(GGML_TYPE_F32, E, T, 1, 1, "inp_embd") : nullptr;
        in[j].hidd = j == 0 && !dev_first ? g.new_tensor(GGML_TYPE_F32, E, T, 1, 1, "inp_hidd") : nullptr;
        in[j].pos  = nullptr;                                                     // a view of pos_all, below
        in[j].mask = g.new_tensor(GGML_TYPE_F32, n_kv, Tpad, 1, 1, "KQ_mask");
        in[j].out  = g.new_tensor(GGML_TYPE_I32, 1, 1, 1, 1, "inp_out_ids");
        for (ggml_tensor * t : { in[j].embd, in[j].hidd, in[j].pos, in[j].mask, in[j].out }) if (t) { t->flags |= GGML_TENSOR_FLAG_INPUT; inputs.push_back(t); }
    }
    // the positions of all steps in ONE input, every step rotating by its slice: the plugin then builds the RoPE table of the chain once
    // instead of computing cos / sin in five q|k epilogues (csrc/graph.cpp)
    ggml_tensor * pos_all = g.new_tensor(GGML_TYPE_I32, T0 + n_steps - 1, 1, 1, 1, "inp_pos");
    pos_all->flags |= GGML_TENSOR_FLAG_INPUT; inputs.push_back(pos_all);
    for (int j = 0; j < n_steps; ++j) in[j].pos = g.view_1d(pos_all, in[j].T, (size_t)(j == 0 ? 0 : T0 + j - 1) * 4);
    const bool saved_want = want_logits;
    want_logits = false;                                        // every step ends in GGML_OP_ARGMAX
    std::vector<ggml_tensor *> amax(n_steps);
    ggml_tensor * prev_norm = nullptr;
    ggml_tensor * dev_hidd = nullptr, * dev_tok = nullptr;
    if (dev_first) {      // leaves over the target's memory (pre-set data: the allocator leaves them alone)
        dev_hidd = g.new_tensor(GGML_TYPE_F32, E, T0, 1, 1, "inp_hidd"); dev_hidd->data = dfeat.data; dev_hidd->buffer = dfeat.buffer;
        dev_tok  = g.new_tensor(GGML_TYPE_I32, T0, 1, 1, 1, "inp_tokens"); dev_tok->data = dids.data; dev_tok->buffer = dids.buffer;
    }
    for (int j = 0; j < n_steps; ++j) {
        ggml_tensor * embd = in[j].embd, * hidd = in[j].hidd;
        if (j == 0 && dev_first) { embd = g.get_rows(tok_embd_dev, dev_tok); hidd = dev_hidd; }
        if (j > 0) { embd = g.get_rows(tok_embd_dev, amax[j - 1]); hidd = prev_norm; }     // the hand-off never leaves the device
        StepIO io{ embd, hidd, in[j].pos, in[j].mask, in[j].out, in[j].T, 1, n_kv, heads[j] };
        ggml_tensor * rn = nullptr, * ro = nullptr, * ra = nullptr;
        build_forward(g, io, false, nullptr, rn, ro, ra);
        amax[j] = ra; prev_norm = rn;
    }
    want_logits = saved_want;
    last_n_nodes = (int) g.nodes.size();
    if (!g.alloc()) { kv = kv_saved; chain_ids = nullptr; return -3; }
    for (int j = 0; j < n_steps; ++j) { amax[j]->data = (char *) chain_ids->data + 4*(j + 1); amax[j]->buffer = chain_ids->buffer; }      // (the slots the allocator gave them stay unused)
    const double t1 = now_us();

    // ---- host image of the inputs, one asynchronous copy
    ggml_tensor * base = inputs[0];
    size_t span = 0; bool packed = true;
    for (ggml_tensor * t : inputs) {
        const ptrdiff_t off = (char *) t->data - (char *) base->data;
        if (t->buffer != base->buffer || off < 0 || (size_t) off > ((size_t) 64 << 20)) { packed = false; break; }
        span = std::max(span, (size_t) off + mh::nbytes(t));
    }
    if (packed && span > stage_cap) {
        if (stage_in) { be->synchronize(); be->host_free(stage_in); }
        stage_cap = span + span/2 + 4096; stage_in = (char *) be->host_alloc(stage_cap);
    }
    static thread_local std::vector<char> unpacked;
    auto host_of = [&](ggml_tensor * t) -> char * { if (packed) return stage_in + ((char *) t->data - (char *) base->data); unpacked.resize(mh::nbytes(t)); return unpacked.data(); };
    auto flush = [&](ggml_tensor * t, char * h) { if (!packed) g.set(t, h, 0, mh::nbytes(t)); };
    { char * h = host_of(chain_ids); memset(h, 0, mh::nbytes(chain_ids)); memcpy(h, &first.token[T0 - 1], 4); flush(chain_ids, h); }
    std::vector<int32_t> pos_host((size_t) T0 + n_steps - 1);
    for (int j = 0; j < n_steps; ++j) {
        const Batch & b = bs[j]; const int T = in[j].T;
        if (in[j].embd) {
            float * embd = (float *) host_of(in[j].embd);
            for (int i = 0; i < T; ++i) { const uint16_t * src = tok_embd.data() + (size_t) b.token[i] * E; h2f_row(src, embd + (size_t) i * E, E); }
            flush(in[j].embd, (char *) embd);
            char * h = host_of(in[j].hidd); memcpy(h, b.hidd.data(), (size_t) T * E * 4); flush(in[j].hidd, h);
        }
        memcpy(pos_host.data() + (j == 0 ? 0 : T0 + j - 1), b.pos.data(), (size_t) T * 4);
        { int32_t last = T - 1; char * h = host_of(in[j].out); memcpy(h, &last, 4); flush(in[j].out, h); }
        float * mask = (float *) host_of(in[j].mask);
        std::fill(mask, mask + (size_t) n_kv * T, -INFINITY);
        for (int r = 0; r < T; ++r) {
            const uint64_t sbit = 1ull << b.seq_first[r]; const int32_t pos = b.pos[r];
            float * row = mask + (size_t) r * n_kv;
            for (int i = 0; i < n_kv; ++i) if ((kv.cells[i].seqs & sbit) && kv.cells[i].pos <= pos) row[i] = 0.0f;
        }
        flush(in[j].mask, (char *) mask);
    }
    { char * h = host_of(pos_all); memcpy(h, pos_host.data(), pos_host.size() * 4); flush(pos_all, h); }
    if (packed) g.set_async(base, stage_in, 0, span);
    const double t2 = now_us();

    enum ggml_status st = g.compute_async();
    const double t3a = now_us();
    if (ids_stage.size() < (size_t) n_steps) ids_stage.resize((size_t) n_steps + 64);
    if (st == GGML_STATUS_SUCCESS) g.get_async(chain_ids, ids_stage.data(), 4, (size_t) n_steps * 4);      // one copy: the steps' slots are contiguous
    g.t_issue_us += t3a - t2;
    t_build_us += t1 - t0; t_upload_us += t2 - t1;
    if (st != GGML_STATUS_SUCCESS) { be->synchronize(); kv = kv_saved; return st == GGML_STATUS_ABORTED ? 2 : -4; }
    chain_steps = n_steps; chain_t_launch = t2;
    if (defer_wait) return 0;                                   // the caller overlaps host work, then collects with chain_wait()
    return chain_wait(ids);
}
int Model::chain_wait(std::vector<int32_t> & ids) {
    if (chain_steps <= 0) return -1;
    const double t3a = now_us();
    be->synchronize();
    const double t3 = now_us();
    gctx->t_wait_us += t3 - t3a;
    const int32_t * p = (const int32_t *) ids_stage.data();
    ids.assign(p, p + chain_steps);
    logits.clear(); hidden.clear(); argmax_ids.clear(); n_outputs = 0; out_ids.clear();
    t_compute_us += t3 - chain_t_launch; n_decode++;
    chain_steps = 0;
    return 0;
}

const float * Model::logits_ith(int i) const { for (int r = 0; r < n_outputs; ++r) if (out_ids[r] == i) return logits.data() + (size_t) r * cfg.n_vocab; return nullptr; }
const float * Model::hidden_ith(int i) const { for (int r = 0; r < n_outputs; ++r) if (out_ids[r] == i) return hidden.data() + (size_t) r * cfg.n_embd; return nullptr; }

} // namespace eh
