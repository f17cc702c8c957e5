/*
 * CPU oracle, C restatement (OpenMP) of the sparse paged fusion attention chain.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py; nothing in the product library links or loads it.  It mirrors oracle/ref.py function by
 * function (which is pinned against golden vectors of the reference's tests/utils.py) and exists so
 * that full-size inputs can be checked and the CPU path can be timed on the GPU box's host cores.
 *
 *   spfa_oracle_cent_select : reference tests/utils.py:87-123 (scores + top-k) and :4-84 (page
 *                             selection), kernel variants cent_select.cpp:218,350-362,410-414
 *   spfa_oracle_attention   : restates the device arithmetic of
 *                             sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:256-294 (gather),
 *                             :3363 (scale), :3408-3419 (online softmax), :3987-3999 (rescale, divide),
 *                             :2647,4160-4174 (int8 KV); attention numerics are "parity unpinned"
 *                             (the reference has no CPU attention).
 *
 * Build: gcc -O3 -fopenmp -mavx2 -mf16c -mfma -shared -fPIC oracle.c -o liboracle.so
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SENTINEL 0x7fffffff
enum { DT_F16 = 0, DT_BF16 = 1, DT_I8 = 2 };

static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }
static inline float b2f(uint16_t b) {
    union { uint32_t u; float f; } x;
    x.u = (uint32_t)b << 16;
    return x.f;
}
static inline float elem16(const void* p, int dt, long i) {
    const uint16_t v = ((const uint16_t*)p)[i];
    return dt == DT_BF16 ? b2f(v) : h2f(v);
}
/* n (multiple of 8) elements of a KV row -> float */
static inline void row_to_f32(const void* row, int dt, int n, float* out) {
    if (dt == DT_F16) {
        const uint16_t* r = (const uint16_t*)row;
        for (int i = 0; i < n; i += 8)
            _mm256_storeu_ps(out + i, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(r + i))));
    } else if (dt == DT_BF16) {
        const uint16_t* r = (const uint16_t*)row;
        for (int i = 0; i < n; ++i) out[i] = b2f(r[i]);
    } else {
        const int8_t* r = (const int8_t*)row;
        for (int i = 0; i < n; ++i) out[i] = (float)r[i];
    }
}

int spfa_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

typedef struct { double s; int i; } ScoreIdx;
static int cmp_desc(const void* a, const void* b) {
    const ScoreIdx* x = (const ScoreIdx*)a;
    const ScoreIdx* y = (const ScoreIdx*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return x->i - y->i; /* equal scores: ascending cluster id */
}

/*
 * q [Bq,Hq,D], l1 [Hkv,C,D] (16-bit, qk_dt), block_ids [Hkv,kv_page_len], block_table [*,max_page],
 * seq_len [B].  Outputs: indices [B,Hq,k] (may be NULL), pos [B,Hq,P], len8 [B,Hq,8], max8 [B,8].
 */
int spfa_oracle_cent_select(const void* q, const void* l1, int qk_dt, const int32_t* block_ids,
                            const int32_t* block_table, const int32_t* seq_len, int B, int Hq, int Hkv, int D,
                            int C, int k, int kv_page_len, int max_page, int block_size, int P, int sink_recent,
                            int cap_pages, int32_t* indices, int32_t* pos, int32_t* len8, int64_t* max8) {
    const int G = Hq / Hkv;
    if (max8) memset(max8, 0, (size_t)B * 8 * sizeof(int64_t));
    int* lens = (int*)calloc((size_t)B * Hq, sizeof(int));
#pragma omp parallel
    {
        ScoreIdx* sc = (ScoreIdx*)malloc((size_t)C * sizeof(ScoreIdx));
        float* qf = (float*)malloc((size_t)D * sizeof(float));
        uint8_t* member = (uint8_t*)malloc((size_t)C);
#pragma omp for collapse(2) schedule(dynamic)
        for (int b = 0; b < B; ++b) {
            for (int h = 0; h < Hq; ++h) {
                const int n2 = h / G;
                for (int d = 0; d < D; ++d) qf[d] = elem16(q, qk_dt, ((long)b * Hq + h) * D + d);
                for (int c = 0; c < C; ++c) {
                    double acc = 0.0; /* exact for fp16 products, rounded once to fp32 like oracle/ref.py */
                    const long base = ((long)n2 * C + c) * D;
                    for (int d = 0; d < D; ++d) acc += (double)qf[d] * (double)elem16(l1, qk_dt, base + d);
                    sc[c].s = (double)(float)acc;
                    sc[c].i = c;
                }
                qsort(sc, (size_t)C, sizeof(ScoreIdx), cmp_desc);
                memset(member, 0, (size_t)C);
                for (int j = 0; j < k; ++j) {
                    member[sc[j].i] = 1;
                    if (indices) indices[((long)b * Hq + h) * k + j] = sc[j].i;
                }
                if (!pos) continue;
                int pl = (seq_len[b] + block_size - 1) / block_size; /* tests/utils.py:44 */
                if (pl > max_page) pl = max_page;
                if (pl < 0) pl = 0;
                const int32_t* bt = block_table + (long)b * max_page;
                int32_t* prow = pos + ((long)b * Hq + h) * P;
                int count = 0;
                for (int i = 0; i < pl; ++i) {
                    int pg = bt[i];
                    if (pg < 0) pg = 0;
                    if (pg > kv_page_len - 1) pg = kv_page_len - 1; /* tests/utils.py:62 */
                    const int cid = block_ids[(long)n2 * kv_page_len + pg];
                    int hit = cid >= 0 && cid < C && member[cid];
                    if (sink_recent) /* cent_select.cpp:350-362 */
                        hit = hit || bt[i] == bt[0] || bt[i] == bt[pl - 1] || bt[i] == bt[pl >= 2 ? pl - 2 : 0];
                    if (hit) {
                        if (count < P) prow[count] = i;
                        ++count;
                    }
                }
                for (int s = count < P ? count : P; s < P; ++s) prow[s] = SENTINEL;
                int len = count < P ? count : P;
                if (cap_pages && len > pl / 8) len = pl / 8; /* cent_select.cpp:218,410-414 */
                lens[(long)b * Hq + h] = len;
                if (len8)
                    for (int j = 0; j < 8; ++j) len8[((long)b * Hq + h) * 8 + j] = len;
            }
        }
        free(sc);
        free(qf);
        free(member);
    }
    if (max8)
        for (int b = 0; b < B; ++b) {
            int m = 0;
            for (int h = 0; h < Hq; ++h)
                if (lens[(long)b * Hq + h] > m) m = lens[(long)b * Hq + h];
            for (int j = 0; j < 8; ++j) max8[(long)b * 8 + j] = (int64_t)m * block_size; /* cent_select.cpp:143-151 */
        }
    free(lens);
    return 0;
}

/*
 * Attention over page lists.  pos [B,rows,P] (rows = Hq or Hkv); row_len (optional, [B,rows]): number of
 * leading slots to use (else: up to the first sentinel); act_seq (optional, [B]): virtual sequence length;
 * token_limit (optional, [B]): logical tokens >= limit are masked.  antiq_* : [2,Hkv,D] 16-bit in qk_dt.
 * out: float [B,Hq,D].
 */
int spfa_oracle_attention(const void* q, int qk_dt, const void* kcache, const void* vcache, int kv_dt,
                          long blk_stride, long tok_stride, long head_stride, int n_blocks,
                          const int32_t* block_table, int max_page, const int32_t* pos, int rows, int P,
                          const int32_t* row_len, const int64_t* act_seq, const int32_t* token_limit,
                          const void* antiq_scale, const void* antiq_offset, int B, int Hq, int Hkv, int D,
                          int block_size, float scale, float* out) {
    const int G = Hq / Hkv;
    const int quant = kv_dt == DT_I8;
#pragma omp parallel
    {
        float* qv = (float*)malloc((size_t)D * sizeof(float));
        float* kf = (float*)malloc((size_t)D * sizeof(float));
        float* o = (float*)malloc((size_t)D * sizeof(float));
        float* s = (float*)malloc((size_t)block_size * sizeof(float));
#pragma omp for collapse(2) schedule(dynamic)
        for (int b = 0; b < B; ++b) {
            for (int h = 0; h < Hq; ++h) {
                const int n2 = h / G;
                const int r = rows == Hq ? h : n2;
                const int32_t* prow = pos + ((long)b * rows + r) * P;
                int n_pages = P;
                if (row_len) n_pages = row_len[(long)b * rows + r];
                for (int j = 0; j < n_pages; ++j)
                    if (prow[j] == SENTINEL || prow[j] < 0 || prow[j] >= max_page) { n_pages = j; break; }
                long n_tok = (long)n_pages * block_size;
                if (act_seq) {
                    if (act_seq[b] < n_tok) n_tok = act_seq[b];
                    const int need = (int)((n_tok + block_size - 1) / block_size);
                    if (need < n_pages) n_pages = need;
                }
                for (int d = 0; d < D; ++d) {
                    qv[d] = elem16(q, qk_dt, ((long)b * Hq + h) * D + d);
                    if (quant) qv[d] *= elem16(antiq_scale, qk_dt, (long)n2 * D + d); /* …Us2.h:2647 */
                    o[d] = 0.f;
                }
                float m = -2e38f, l = 0.f; /* …Us2.h:859,4472 */
                for (int j = 0; j < n_pages; ++j) {
                    const int p = prow[j];
                    int blk = block_table[(long)b * max_page + p];
                    if (blk < 0) blk = 0;
                    if (blk > n_blocks - 1) blk = n_blocks - 1;
                    long valid = n_tok - (long)j * block_size;
                    if (valid > block_size) valid = block_size;
                    if (token_limit) {
                        const long lim = (long)token_limit[b] - (long)p * block_size;
                        if (lim < valid) valid = lim;
                    }
                    if (valid <= 0) continue;
                    const char* kp = (const char*)kcache + (long)blk * blk_stride + (long)n2 * head_stride;
                    const char* vp = (const char*)vcache + (long)blk * blk_stride + (long)n2 * head_stride;
                    float mx = m;
                    for (int t = 0; t < valid; ++t) {
                        row_to_f32(kp + (long)t * tok_stride, kv_dt, D, kf);
                        float acc = 0.f;
#pragma omp simd reduction(+ : acc)
                        for (int d = 0; d < D; ++d) acc += qv[d] * kf[d];
                        s[t] = acc * scale; /* …Us2.h:3363 */
                        if (s[t] > mx) mx = s[t];
                    }
                    const float alpha = expf(m - mx);
                    l *= alpha;
                    for (int d = 0; d < D; ++d) o[d] *= alpha;
                    for (int t = 0; t < valid; ++t) {
                        const float pt = expf(s[t] - mx);
                        l += pt;
                        row_to_f32(vp + (long)t * tok_stride, kv_dt, D, kf);
#pragma omp simd
                        for (int d = 0; d < D; ++d) o[d] += pt * kf[d];
                    }
                    m = mx;
                }
                float* orow = out + ((long)b * Hq + h) * D;
                for (int d = 0; d < D; ++d) {
                    float v = l > 0.f ? o[d] / l : 0.f;
                    if (quant && l > 0.f) { /* …Us2.h:4160-4174 */
                        const long pi = (long)Hkv * D + (long)n2 * D + d;
                        if (antiq_offset) v += elem16(antiq_offset, qk_dt, pi);
                        v *= elem16(antiq_scale, qk_dt, pi);
                    }
                    orow[d] = v;
                }
            }
        }
        free(qv);
        free(kf);
        free(o);
        free(s);
    }
    return 0;
}
