// psgpu_lm.hip -- the trigram language model on the device: table upload and the
// batch look-up entry point.  The look-up itself is psgpu_lm_dev.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <vector>

#include "psgpu.h"
#include "psgpu_internal.h"
#include "psgpu_lm_dev.h"

struct psgpu_lm_s {
    LmDev d;
    LmDev *d_dev = nullptr;          // the same descriptor in device memory (lm_tg_score_call)
    std::vector<void *> allocs;
};

// internal: the searches copy the descriptor (psgpu_fwdtree_set_lm / psgpu_fwdflat_set_lm) or use its device copy
extern "C" const LmDev *psgpu_lm_dev(const psgpu_lm_t *lm) { return lm ? &lm->d : nullptr; }
extern "C" const LmDev *psgpu_lm_dev_ptr(const psgpu_lm_t *lm) { return lm ? lm->d_dev : nullptr; }

static int psgpu_fail(int code, const char *msg) { psgpu_set_error("%s", msg); return code; }

static const void *lm_up(psgpu_lm_s *m, const void *src, size_t n, size_t pad, int *rc)
{
    void *d = nullptr;
    if (*rc != PSGPU_OK) return nullptr;
    if (hipMalloc(&d, n + pad ? n + pad : 4) != hipSuccess) { *rc = psgpu_fail(PSGPU_EHIP, "psgpu_lm_create: hipMalloc failed"); return nullptr; }
    m->allocs.push_back(d);
    if (pad && hipMemset((char *)d + n, 0, pad) != hipSuccess) { *rc = psgpu_fail(PSGPU_EHIP, "psgpu_lm_create: hipMemset failed"); return nullptr; }
    if (n && hipMemcpy(d, src, n, hipMemcpyHostToDevice) != hipSuccess) { *rc = psgpu_fail(PSGPU_EHIP, "psgpu_lm_create: hipMemcpy failed"); return nullptr; }
    return d;
}

extern "C" int psgpu_lm_create(psgpu_lm_t **out, const psgpu_lm_tables_t *t)
{
    if (!out || !t || !t->unigrams || !t->widmap) return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_create: NULL argument");
    if (t->order < 1 || t->order > PSGPU_LM_MAX_LEVELS + 1) return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_create: order must be 1..5");
    if (t->order > 1 && (!t->ngram_mem || !t->quant)) return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_create: the trie arrays are missing");
    for (int l = 0; l < t->order - 1; ++l)
        if (t->word_bits[l] > 25 || t->next_bits[l] > 25 || t->level_offset[l] > t->ngram_mem_size)
            return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_create: a level's bit layout is outside what bitarr_read_int25 can read");
    for (int32_t w = 0; w < t->n_words; ++w)
        if (t->widmap[w] < -1 || t->widmap[w] >= t->n_unigrams || (t->histmap && (t->histmap[w] < -1 || t->histmap[w] >= t->n_unigrams)))
            return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_create: widmap entry outside the model's unigrams");
    {
        const int dev = psgpu_check_device();
        if (dev != PSGPU_OK) return dev;
    }
    psgpu_lm_s *m = new psgpu_lm_s();
    int rc = PSGPU_OK;
    LmDev &d = m->d;
    d.order = t->order; d.n_unigrams = t->n_unigrams; d.n_words = t->n_words;
    d.ug = (const uint32_t *)lm_up(m, t->unigrams, 12 * ((size_t)t->n_unigrams + 1), 0, &rc);
    d.mem = (const uint8_t *)lm_up(m, t->ngram_mem, t->order > 1 ? (size_t)t->ngram_mem_size : 0, 16, &rc);
    d.quant = (const float *)lm_up(m, t->quant, t->order > 1 ? (size_t)(2 * (t->order - 2) + 1) * 65536 * 4 : 0, 0, &rc);
    d.widmap = (const int32_t *)lm_up(m, t->widmap, 4 * (size_t)t->n_words, 0, &rc);
    d.cwt = t->class_weight ? (const int32_t *)lm_up(m, t->class_weight, 4 * (size_t)t->n_words, 0, &rc) : nullptr;
    d.histmap = t->histmap ? (const int32_t *)lm_up(m, t->histmap, 4 * (size_t)t->n_words, 0, &rc) : d.widmap;
    d.n_set = 0; d.set = nullptr; d.set_lw = nullptr; d.addtab = nullptr; d.addtab_n = 0; d.add_zero = 0;
    for (int l = 0; l < t->order - 1; ++l) {
        d.lev[l].off = t->level_offset[l]; d.lev[l].total_bits = t->total_bits[l]; d.lev[l].word_bits = t->word_bits[l];
        d.lev[l].word_mask = t->word_mask[l]; d.lev[l].max_vocab = t->max_vocab[l]; d.lev[l].next_bits = t->next_bits[l];
        d.lev[l].next_mask = t->next_mask[l];
    }
    d.lw = t->lw; d.log_wip = t->log_wip; d.log_zero = t->log_zero;
    if (rc == PSGPU_OK) m->d_dev = (LmDev *)const_cast<void *>(lm_up(m, &d, sizeof d, 0, &rc));
    if (rc != PSGPU_OK) { psgpu_lm_free(m); return rc; }
    *out = m;
    return PSGPU_OK;
}

extern "C" int psgpu_lm_create_interp(psgpu_lm_t **out, const psgpu_lm_t *const *members, const int32_t *lweights, int32_t n_members,
                                      const void *addtab, int32_t width, int32_t addtab_size, int32_t add_zero, int32_t log_zero)
{
    if (!out || !members || !lweights || n_members < 1 || n_members > 64) return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_create_interp: 1..64 members");
    if (addtab_size < 0 || (addtab_size > 0 && (!addtab || (width != 1 && width != 2 && width != 4))))
        return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_create_interp: the log-add table is [size] of 1, 2 or 4 bytes");
    int order = 0;
    for (int i = 0; i < n_members; ++i) {
        if (!members[i] || members[i]->d.n_set || members[i]->d.n_words != members[0]->d.n_words)
            return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_create_interp: members are plain models over one word list");
        order = std::max(order, members[i]->d.order);
    }
    psgpu_lm_s *m = new psgpu_lm_s();
    int rc = PSGPU_OK;
    std::vector<LmDev> mem((size_t)n_members);
    for (int i = 0; i < n_members; ++i) mem[i] = members[i]->d;
    std::vector<uint32_t> tab((size_t)std::max(addtab_size, 1), 0u);
    for (int i = 0; i < addtab_size; ++i)
        tab[i] = width == 1 ? ((const uint8_t *)addtab)[i] : width == 2 ? ((const uint16_t *)addtab)[i] : ((const uint32_t *)addtab)[i];
    LmDev &d = m->d;
    memset(&d, 0, sizeof d);
    d.order = order; d.n_words = members[0]->d.n_words; d.n_unigrams = 0; d.log_zero = log_zero; d.lw = 1.0f;
    d.n_set = n_members;
    d.set = (const LmDev *)lm_up(m, mem.data(), sizeof(LmDev) * mem.size(), 0, &rc);
    d.set_lw = (const int32_t *)lm_up(m, lweights, 4 * (size_t)n_members, 0, &rc);
    d.addtab = (const uint32_t *)lm_up(m, tab.data(), 4 * tab.size(), 0, &rc);
    d.addtab_n = addtab_size; d.add_zero = add_zero;
    if (rc == PSGPU_OK) m->d_dev = (LmDev *)const_cast<void *>(lm_up(m, &d, sizeof d, 0, &rc));
    if (rc != PSGPU_OK) { psgpu_lm_free(m); return rc; }
    *out = m;
    return PSGPU_OK;
}

extern "C" void psgpu_lm_free(psgpu_lm_t *m)
{
    if (!m) return;
    for (void *p : m->allocs) (void)hipFree(p);
    delete m;
}

__global__ void __launch_bounds__(256)
lm_tg_score_kernel(LmDev m, const int32_t *__restrict__ w3, const int32_t *__restrict__ w2, const int32_t *__restrict__ w1,
                   int64_t n, int32_t *__restrict__ score, int32_t *__restrict__ n_used)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int nu;
        const int32_t a = w3[i];
        // a word id outside the dictionary is the caller's error; the reference would read outside widmap
        const int32_t s = (a < 0 || a >= m.n_words || w2[i] >= m.n_words || w1[i] >= m.n_words)
            ? (nu = 0, m.log_zero) : lm_tg_score(m, a, w2[i], w1[i], nu);
        score[i] = s;
        if (n_used) n_used[i] = nu;
    }
}

extern "C" int psgpu_lm_tg_score_dev(const psgpu_lm_t *lm, const int32_t *w3_dev, const int32_t *w2_dev, const int32_t *w1_dev,
                                     int64_t n, int32_t *score_dev, int32_t *n_used_dev, void *stream)
{
    if (!lm || n < 0) return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_tg_score_dev: bad argument");
    if (n == 0) return PSGPU_OK;
    if (!w3_dev || !w2_dev || !w1_dev || !score_dev) return psgpu_fail(PSGPU_EINVAL, "psgpu_lm_tg_score_dev: NULL device buffer");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(lm_tg_score_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       lm->d, w3_dev, w2_dev, w1_dev, n, score_dev, n_used_dev);
    if (hipGetLastError() != hipSuccess) return psgpu_fail(PSGPU_EHIP, "psgpu_lm_tg_score_dev: launch failed");
    return PSGPU_OK;
}
