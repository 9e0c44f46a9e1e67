/* textcnn_ref.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * Scalar C restatement of the TextCNN encoder path and of its max-pool-sparse backward, written
 * from the formulas, independent of torch:
 *   forward : models/deepconn/layers.py:22-24 (embedding), utils.py:49-61 (masked_fill),
 *             layers.py:46-60 ('same' / 'valid' cross-correlation per width, width-major channels),
 *             layers.py:107-109 (ReLU or tanh, max over the pooled positions, FIRST maximum wins)
 *   backward: what loss.backward() (trainer/train_deepconn_pp.py:165) yields for these ops, computed
 *             through the single conv window per (doc, channel) that the max-pool selects.
 * Pinned by tests/test_c_oracle.py against oracle/ref_cpu.py (itself pinned to the reference's golden
 * vectors) -- forward values, argmax, and autograd's dense gradients.
 * Build: make -C oracle   ->  oracle/_build/libtextcnn_ref.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float act_f(int act, float y) { return act == 0 ? (y > 0.f ? y : 0.f) : tanhf(y); }
static float act_g(int act, float f, float d) { return act == 0 ? (f > 0.f ? d : 0.f) : d * (1.f - f * f); }

/* x[doc, p, :] = mask ? gate * table[ids] : 0 ; returns pointer to the table row or NULL */
static const float* row_of(const int64_t* ids, const uint8_t* mask, const float* table, int D, int L, int doc, int p) {
    if (p < 0 || p >= L) return NULL;
    if (mask && !mask[(size_t)doc * L + p]) return NULL;
    return table + (size_t)ids[(size_t)doc * L + p] * D;
}

/* W[w]: [ch[w], D, kz[w]]  bias[w]: [ch[w]];  feat/argmax: [n_docs, C] */
void textcnn_fwd_ref(int n_docs, int L, int D, int n_widths, const int* kz, const int* ch, int pad_valid, int act,
                     const int64_t* ids, const uint8_t* mask, const float* gate, const float* table,
                     const float* const* W, const float* const* bias, float* feat, int32_t* argmax) {
    int C = 0;
    for (int w = 0; w < n_widths; ++w) C += ch[w];
    for (int doc = 0; doc < n_docs; ++doc) {
        int c0 = 0;
        for (int w = 0; w < n_widths; ++w) {
            const int k = kz[w], pad = pad_valid ? 0 : (k - 1) / 2, npos = pad_valid ? L - k + 1 : L;
            for (int c = 0; c < ch[w]; ++c) {
                float best = -INFINITY;
                int bidx = 0;
                for (int l = 0; l < npos; ++l) {
                    float y = bias[w][c];
                    for (int j = 0; j < k; ++j) {
                        const float* r = row_of(ids, mask, table, D, L, doc, l + j - pad);
                        if (!r) continue;
                        const float gv = gate ? gate[(size_t)doc * L + l + j - pad] : 1.f;
                        for (int d = 0; d < D; ++d) y += gv * r[d] * W[w][((size_t)c * D + d) * k + j];
                    }
                    if (y > best) { best = y; bidx = l; }
                }
                feat[(size_t)doc * C + c0 + c] = act_f(act, best);
                argmax[(size_t)doc * C + c0 + c] = bidx;
            }
            c0 += ch[w];
        }
    }
}

/* dW/dbias overwritten; dtable [V, D] and dgate [n_docs, L] accumulated (may be NULL) */
void textcnn_bwd_sparse_ref(int n_docs, int L, int D, int V, int n_widths, const int* kz, const int* ch, int pad_valid,
                            int act, int padding_idx, const int64_t* ids, const uint8_t* mask, const float* gate,
                            const float* table, const float* const* W, const float* feat, const int32_t* argmax,
                            const float* d_feat, float* const* dW, float* const* dbias, float* dtable, float* dgate) {
    (void)V;
    int C = 0;
    for (int w = 0; w < n_widths; ++w) C += ch[w];
    for (int w = 0; w < n_widths; ++w) {
        memset(dW[w], 0, sizeof(float) * (size_t)ch[w] * D * kz[w]);
        memset(dbias[w], 0, sizeof(float) * (size_t)ch[w]);
    }
    for (int doc = 0; doc < n_docs; ++doc) {
        int c0 = 0;
        for (int w = 0; w < n_widths; ++w) {
            const int k = kz[w], pad = pad_valid ? 0 : (k - 1) / 2;
            for (int c = 0; c < ch[w]; ++c) {
                const size_t o = (size_t)doc * C + c0 + c;
                const float g = act_g(act, feat[o], d_feat[o]);
                if (g == 0.f) continue;
                dbias[w][c] += g;
                for (int j = 0; j < k; ++j) {
                    const int p = argmax[o] + j - pad;
                    const float* r = row_of(ids, mask, table, D, L, doc, p);
                    if (!r) continue;
                    const size_t tok = (size_t)doc * L + p;
                    const float gv = gate ? gate[tok] : 1.f;
                    const int64_t id = ids[tok];
                    float dot = 0.f;
                    for (int d = 0; d < D; ++d) {
                        const float wv = W[w][((size_t)c * D + d) * k + j];
                        dW[w][((size_t)c * D + d) * k + j] += g * gv * r[d];
                        if (dtable && id != padding_idx) dtable[(size_t)id * D + d] += g * gv * wv;
                        dot += g * wv * r[d];
                    }
                    if (dgate) dgate[tok] += dot;
                }
            }
            c0 += ch[w];
        }
    }
}
