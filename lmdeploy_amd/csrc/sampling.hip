// Stochastic sampling: temperature -> top-k -> softmax -> top-p -> min-p -> draw, one token per sequence.
//
// Replaces: Sampling::Forward (src/turbomind/generation/sampling.cc:92-183) = invokeTopKSortFilter / invokeSoftmax +
// invokeTopPSort (kernels/sampling_topk_kernels.cu, sampling_topp_kernels.cu), invokeTopPMinPFilter
// (sampling_topp_kernels.cu:291-384), invokeSampling (sampling_kernels.cu:16-94), and the temperature step of the logits
// processor (generation/logits_processor.cc:105-112).  Semantics kept: the candidates are ordered by descending
// probability (ties: lower token id first), top-k keeps the first k, top-p keeps the shortest prefix whose cumulative
// probability EXCEEDS top_p, min-p drops candidates below min_p * p_max, the survivors are renormalised and the token
// is the first one whose inclusive prefix sum exceeds the uniform draw.
//
// MI355X design: no sort.  The logits are fp16, so a row has at most 65536 distinct values and all tokens with the
// same fp16 logit have the same probability.  Pass A builds an INTEGER histogram over the order-preserving 16-bit key
// of each logit (global atomics, deterministic); pass B walks the 65536 bins in descending order with fp64
// accumulators (count x exp((v - vmax)/T)): top-k cut, normaliser, top-p cut, min-p cut and the drawn (bin, ordinal)
// all fall out of that walk; pass C finds the ordinal-th token (in id order) carrying the drawn fp16 value.
// 3 passes over a 256 KB row instead of a 128k-element segmented sort per row and step.  The uniform draw comes from
// Philox4x32-10 keyed by the request's seed with the context length as counter (curand's XORWOW stream cannot be
// reproduced, so sampling parity is defined on the filtered distribution + the draw-to-token mapping, not on curand).
#include "tm_common.h"
#include "tm_kernels.h"

namespace tmk {

constexpr int kBins = 65536;

// fp16 bit pattern -> key that sorts like the value (NaN lowest)
__device__ __forceinline__ uint32_t key_of(uint16_t h)
{
    if ((h & 0x7c00u) == 0x7c00u && (h & 0x03ffu)) {
        return 0u;  // NaN
    }
    if (h == 0x8000u) {
        return 0x8000u;  // -0.0 sorts with +0.0 (equal values: the tie goes by token id, as in a stable sort on the float value)
    }
    return (h & 0x8000u) ? (uint32_t)(uint16_t)~h : (uint32_t)(h | 0x8000u);
}

__device__ __forceinline__ float value_of_key(uint32_t key)
{
    const uint16_t h = (key & 0x8000u) ? (uint16_t)(key & 0x7fffu) : (uint16_t)~key;
    return (float)bit_cast<half_t>(h);
}

// ---- pass A: integer histogram of the row ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_hist_kernel(uint32_t* __restrict__ hist, const half_t* __restrict__ logits,
                                                          int V, int ld)
{
    const int       b   = blockIdx.y;
    const uint16_t* row = (const uint16_t*)(logits + (size_t)b * ld);
    uint32_t*       h   = hist + (size_t)b * kBins;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 8; i < V; i += gridDim.x * 256 * 8) {
        if (i + 8 <= V) {
            const u32x4 w = *(const u32x4*)(row + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(h + key_of((uint16_t)(w[e] & 0xffffu)), 1u);
                atomicAdd(h + key_of((uint16_t)(w[e] >> 16)), 1u);
            }
        }
        else {
            for (int j = i; j < V; ++j) {
                atomicAdd(h + key_of(row[j]), 1u);
            }
        }
    }
}

// block-wide exclusive scans (1024 threads), fixed combination order -> deterministic
template<class T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* smem /*[16]*/, T* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T         x    = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T y = __shfl_up(x, d);
        if (lane >= d) {
            x += y;
        }
    }
    if (lane == 63) {
        smem[wave] = x;
    }
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) {
            base += smem[w];
        }
        tot += smem[w];
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// ---- pass B + C: one 1024-thread workgroup per row ------------------------------------------------------------------
// Thread t owns the 64 bins j = 64 t .. 64 t + 63 of the DESCENDING order (bin j <-> key 65535 - j).
__global__ __launch_bounds__(1024) void sample_select_kernel(int* __restrict__ out_ids,
                                                             uint32_t* __restrict__ hist,
                                                             const half_t* __restrict__ logits,
                                                             int V,
                                                             int ld,
                                                             const float* __restrict__ temperature,
                                                             const int* __restrict__ top_k,
                                                             const float* __restrict__ top_p,
                                                             const float* __restrict__ min_p,
                                                             const float* __restrict__ uniform,
                                                             int* __restrict__ kept_out,
                                                             int keep_hist)
{
    __shared__ double   s_d[16];
    __shared__ long long s_l[16];
    __shared__ int      s_i[8];
    __shared__ double   s_v[4];
    const int  b   = blockIdx.x;
    const int  tid = threadIdx.x;
    uint32_t*  h   = hist + (size_t)b * kBins;
    const float T  = temperature ? temperature[b] : 1.0f;
    const float invT = 1.0f / (T > 0.f ? T : 1.0f);
    long long   k_lim = (top_k && top_k[b] > 0) ? top_k[b] : (long long)V;
    const double tp   = top_p ? (double)top_p[b] : 1.0;
    const double mp   = min_p ? (double)min_p[b] : 0.0;
    const double u    = uniform ? (double)uniform[b] : 0.0;

    // My 64 bins are re-read from the (L2-resident) histogram in every walk instead of living in 64 + 128 registers:
    // a 1024-thread workgroup has 128 VGPRs per lane.  bin(q) = raw count of descending bin 64 tid + q.
    auto bin = [&](int q) -> long long { return (long long)h[kBins - 1 - (tid * 64 + q)]; };
    long long mine  = 0;
    int       first = kBins;  // largest value present = first non-empty bin
    for (int q = 63; q >= 0; --q) {
        const long long c = bin(q);
        mine += c;
        if (c) {
            first = tid * 64 + q;
        }
    }
    if (tid < 8) {
        s_i[tid] = tid == 3 ? -1 : kBins;
    }
    __syncthreads();
    atomicMin(&s_i[0], first);
    __syncthreads();
    const float vmax = value_of_key(kBins - 1 - s_i[0]);

    // top-k: exactly min(k, V) candidates survive (ties inside the cut bin: lowest ids first); every walk below clips
    // the raw counts with the running prefix: kept(q) = clamp(k_lim - run, 0, raw)
    long long       tot_cnt;
    const long long before = block_exclusive_scan<long long>(mine, s_l, &tot_cnt);
    if (k_lim > tot_cnt) {
        k_lim = tot_cnt;
    }
    // weight of one candidate of bin q
    auto wq = [&](int q) -> double {
        const float v = value_of_key(kBins - 1 - (tid * 64 + q));
        const double w = exp((double)((v - vmax) * invT));
        return w == w ? w : 0.0;  // NaN logits carry no probability
    };
    auto clip = [](long long room, long long raw) -> long long { return room <= 0 ? 0 : (room < raw ? room : raw); };

    double msum = 0.0;
    {
        long long run = before;
        for (int q = 0; q < 64; ++q) {
            const long long raw = bin(q);
            const long long c   = clip(k_lim - run, raw);
            run += raw;
            if (c) {
                msum += wq(q) * (double)c;
            }
        }
    }
    double       Z;
    const double mbefore = block_exclusive_scan<double>(msum, s_d, &Z);
    const double pmax    = 1.0 / Z;  // the best bin has weight exp(0) = 1

    // ---- top-p: first candidate (bin j, ordinal m) whose inclusive cumulative probability exceeds top_p -------------
    // kept = candidates up to and including it, ksum = their probability mass (s_kept / s_sum of the reference)
    if (tid == 0) {
        s_l[0] = k_lim;
        s_v[0] = 1.0;
    }
    __syncthreads();
    if (tp < 1.0) {
        double    cum = mbefore / Z;
        long long run = before, c0 = before < k_lim ? before : k_lim;
        int       hit = -1;
        long long hit_kept = 0;
        double    hit_sum  = 0.0;
        for (int q = 0; q < 64 && hit < 0; ++q) {
            const long long raw = bin(q);
            const long long c   = clip(k_lim - run, raw);
            run += raw;
            if (c) {
                const double p   = wq(q) / Z;
                const double end = cum + p * (double)c;
                if (end > tp) {
                    long long m = (long long)floor((tp - cum) / p) + 1;
                    m           = m < 1 ? 1 : (m > c ? c : m);
                    while (m > 1 && cum + p * (double)(m - 1) > tp) {
                        --m;
                    }
                    while (m < c && !(cum + p * (double)m > tp)) {
                        ++m;
                    }
                    hit      = tid * 64 + q;
                    hit_kept = c0 + m;
                    hit_sum  = cum + p * (double)m;
                }
                cum = end;
                c0 += c;
            }
        }
        if (hit >= 0) {
            atomicMin(&s_i[1], hit);
        }
        __syncthreads();
        if (hit >= 0 && s_i[1] == hit) {
            s_l[0] = hit_kept;
            s_v[0] = hit_sum;
        }
        __syncthreads();
    }
    long long kept = s_l[0];
    double    ksum = s_v[0];
    __syncthreads();

    // ---- min-p: candidates with p < min_p * p_max go (they form a suffix of the descending order) -------------------
    if (mp > 0.0) {
        const double thr  = pmax * mp;
        long long    run  = before, c0 = before < kept ? before : kept;
        long long    n_ok = 0;
        double       m_ok = 0.0;
        for (int q = 0; q < 64; ++q) {
            const long long raw  = bin(q);
            const long long c    = clip(k_lim - run, raw);
            const long long take = clip(kept - c0, c);
            run += raw;
            c0 += c;
            if (take > 0) {
                const double p = wq(q) / Z;
                if (p >= thr) {
                    n_ok += take;
                    m_ok += p * (double)take;
                }
            }
        }
        long long tn;
        double    tm;
        (void)block_exclusive_scan<long long>(n_ok, s_l, &tn);
        (void)block_exclusive_scan<double>(m_ok, s_d, &tm);
        kept = tn;
        ksum = tm;
    }

    // ---- draw: first kept candidate whose inclusive prefix sum exceeds u * ksum ---------------------------------------
    {
        const double target = u * ksum;
        double       cum    = mbefore / Z;
        long long    run = before, c0 = before < kept ? before : kept;
        int          hit = -1, hit_m = 0, last_bin = -1, last_m = 0;
        for (int q = 0; q < 64; ++q) {
            const long long raw  = bin(q);
            const long long c    = clip(k_lim - run, raw);
            const long long take = clip(kept - c0, c);
            run += raw;
            c0 += c;
            if (take > 0) {
                const double p   = wq(q) / Z;
                const double end = cum + p * (double)take;
                last_bin         = tid * 64 + q;
                last_m           = (int)take;
                if (hit < 0 && end > target) {
                    long long m = (long long)floor((target - cum) / p) + 1;
                    m           = m < 1 ? 1 : (m > take ? take : m);
                    while (m > 1 && cum + p * (double)(m - 1) > target) {
                        --m;
                    }
                    while (m < take && !(cum + p * (double)m > target)) {
                        ++m;
                    }
                    hit   = tid * 64 + q;
                    hit_m = (int)m;
                }
                cum = end;
            }
        }
        if (hit >= 0) {
            atomicMin(&s_i[2], hit);
        }
        if (last_bin >= 0) {
            atomicMax(&s_i[3], last_bin);  // fallback (rounding: nothing exceeded the target): the last kept candidate
        }
        __syncthreads();
        if (hit >= 0 && s_i[2] == hit) {
            s_i[4] = hit;
            s_i[5] = hit_m;
        }
        __syncthreads();
        if (s_i[2] == kBins && last_bin >= 0 && s_i[3] == last_bin) {
            s_i[4] = last_bin;
            s_i[5] = last_m;
        }
        __syncthreads();
    }
    // No candidate survived, or the distribution is not a distribution (NaN / inf logits: vmax not finite, Z not a positive
    // finite number): the row still gets a DEFINED token -- the arg-max over its finite logits, lowest id on ties, token 0
    // if there is none -- instead of silently keeping the previous step's id.
    const bool degenerate = s_i[4] == kBins || !(Z > 0.0 && Z < 1e300) || !(__builtin_fabsf(vmax) <= 65504.f);
    const uint32_t sel_key = (uint32_t)(kBins - 1 - s_i[4]);
    const int      sel_m   = s_i[5];  // 1-based ordinal among the tokens with that logit, in id order
    if (tid == 0 && kept_out) {
        kept_out[b] = (int)kept;
    }
    // the histogram is left zeroed for the next step (keep_hist: sample_logprobs_kernel reads it first and zeroes it)
    if (!keep_hist) {
        for (int q = 0; q < 64; ++q) {
            h[kBins - 1 - (tid * 64 + q)] = 0u;
        }
    }

    // ---- pass C: the sel_m-th token (ascending id) whose logit has the drawn value ----------------------------------
    const uint16_t* row   = (const uint16_t*)(logits + (size_t)b * ld);
    const int       per   = (V + 1023) / 1024;
    const int       begin = tid * per;
    const int       end   = min(begin + per, V);
    if (degenerate) {
        __shared__ unsigned long long s_best;
        if (tid == 0) {
            s_best = 0ull;
        }
        __syncthreads();
        unsigned long long best = 0ull;
        for (int i = begin; i < end; ++i) {
            const uint16_t bits = row[i];
            if ((bits & 0x7c00u) != 0x7c00u) {  // finite
                const unsigned long long k = ((unsigned long long)(key_of(bits) + 1u) << 32) | (uint32_t)(0x7fffffff - i);
                best                       = k > best ? k : best;
            }
        }
        if (best) {
            atomicMax(&s_best, best);
        }
        __syncthreads();
        if (tid == 0) {
            out_ids[b] = s_best ? 0x7fffffff - (int)(uint32_t)(s_best & 0xffffffffull) : 0;
        }
        return;
    }
    long long       my    = 0;
    for (int i = begin; i < end; ++i) {
        my += key_of(row[i]) == sel_key;
    }
    long long tot;
    const long long prior = block_exclusive_scan<long long>(my, s_l, &tot);
    if (prior < sel_m && sel_m <= prior + my) {
        long long seen = prior;
        for (int i = begin; i < end; ++i) {
            if (key_of(row[i]) == sel_key && ++seen == sel_m) {
                out_ids[b] = i;
                break;
            }
        }
    }
}

// ---- logprobs of the kept candidates (sampling_kernels.cu:67-90 + the Python side's view of it, turbomind.py:472-503) --------
// After sample_select_kernel (keep_hist = 1): the first L = min(kept, cap) candidates of the row in sampling order (descending
// probability, ties by ascending id) with logf of their RENORMALISED probability (the reference's `logits` array holds the kept
// candidates' probabilities divided by their mass when invokeSampling runs), their count, and the drawn token's own logprob.
// force_last (the reference's layout, cap = kMaxLogProb): a drawn token beyond the first `cap` candidates replaces entry cap - 1.
// Records land at row (*step) * step_stride + (row0 + b) * row_stride of the output arrays (step = nullptr: 0); nothing is written
// when that step is >= max_steps.  Structure: thread t owns 64 descending bins like the selection kernel; one pass over the row in id order
// compacts the tokens of the strictly better bins into an LDS list (< cap <= 1024 entries, bitonic sort by (bin, id)) and writes the
// first tokens of the cut bin behind them (already in id order).  Leaves the histogram zeroed.
__global__ __launch_bounds__(1024) void sample_logprobs_kernel(float* __restrict__ out_vals, int* __restrict__ out_idx,
                                                               int* __restrict__ out_num, float* __restrict__ out_sel, int cap,
                                                               int force_last, uint32_t* __restrict__ hist,
                                                               const half_t* __restrict__ logits, int V, int ld,
                                                               const float* __restrict__ temperature,
                                                               const int* __restrict__ kept, const int* __restrict__ sel_ids,
                                                               const int* __restrict__ step, int step_stride, int row_stride,
                                                               int row0, int max_steps)
{
    __shared__ double             s_d[16];
    __shared__ long long          s_l[16];
    __shared__ int                s_i[4];
    __shared__ unsigned long long s_a[1024];
    const int  b   = blockIdx.x;
    const int  tid = threadIdx.x;
    uint32_t*  h   = hist + (size_t)b * kBins;
    const int  st  = step ? *step : 0;
    const bool rec = st < max_steps;
    const size_t row = (size_t)st * step_stride + (size_t)(row0 + b) * row_stride;
    const float T    = temperature ? temperature[b] : 1.0f;
    const float invT = 1.0f / (T > 0.f ? T : 1.0f);
    long long   n    = kept[b];
    const int   sel  = sel_ids[b];

    auto bin = [&](int q) -> long long { return (long long)h[kBins - 1 - (tid * 64 + q)]; };
    long long mine  = 0;
    int       first = kBins;
    for (int q = 63; q >= 0; --q) {
        const long long c = bin(q);
        mine += c;
        if (c) {
            first = tid * 64 + q;
        }
    }
    if (tid < 4) {
        s_i[tid] = tid == 0 ? kBins : -1;
    }
    __syncthreads();
    atomicMin(&s_i[0], first);
    __syncthreads();
    const float vmax = value_of_key(kBins - 1 - min(s_i[0], kBins - 1));
    long long       tot_cnt;
    const long long before = block_exclusive_scan<long long>(mine, s_l, &tot_cnt);
    n = n < 0 ? 0 : (n > tot_cnt ? tot_cnt : n);
    const long long L = n < cap ? n : cap;
    auto wbin = [&](int dbin) -> double {
        const float  v = value_of_key(kBins - 1 - dbin);
        const double w = exp((double)((v - vmax) * invT));
        return w == w ? w : 0.0;
    };
    auto clip = [](long long room, long long raw) -> long long { return room <= 0 ? 0 : (room < raw ? room : raw); };
    // mass of the first n candidates; the cut bin of the first L: prefix P < L <= P + count
    double msum = 0.0;
    {
        long long run = before;
        for (int q = 0; q < 64; ++q) {
            const long long raw = bin(q);
            const long long c   = clip(n - run, raw);
            if (c) {
                msum += wbin(tid * 64 + q) * (double)c;
            }
            if (raw && run < L && L <= run + raw) {
                s_i[1] = tid * 64 + q;
                s_i[2] = (int)run;
            }
            run += raw;
        }
    }
    double S;
    (void)block_exclusive_scan<double>(msum, s_d, &S);
    const bool ok   = L > 0 && S > 0.0 && S < 1e300 && __builtin_fabsf(vmax) <= 65504.f;
    const int  jc   = s_i[1];            // descending index of the cut bin
    const int  nA   = s_i[2];            // candidates in strictly better bins (< L <= 1024)
    const int  nB   = (int)L - nA;       // candidates taken from the cut bin, lowest ids first
    auto lp_of = [&](int dbin) -> float { return logf((float)(wbin(dbin) / S)); };
    for (int q = 0; q < 64; ++q) {       // every thread has read its bins for the last time
        h[kBins - 1 - (tid * 64 + q)] = 0u;
    }
    s_a[tid] = ~0ull;
    if (!ok) {
        if (tid == 0 && rec) {
            out_num[row] = 0;
            out_sel[row] = 0.f;
        }
        return;
    }
    const uint16_t* lrow  = (const uint16_t*)(logits + (size_t)b * ld);
    const int       per   = (V + 1023) / 1024;
    const int       begin = tid * per;
    const int       end   = min(begin + per, V);
    long long cnt = 0;                   // (A count << 32) | B count of my id range
    for (int i = begin; i < end; ++i) {
        const int dbin = kBins - 1 - (int)key_of(lrow[i]);
        cnt += dbin < jc ? (1ll << 32) : (dbin == jc ? 1ll : 0ll);
    }
    long long tot;
    const long long prior = block_exclusive_scan<long long>(cnt, s_l, &tot);
    int pa = (int)(prior >> 32), pb = (int)(prior & 0xffffffffll);
    float*  vals = out_vals + row * (size_t)cap;
    int*    idx  = out_idx + row * (size_t)cap;
    for (int i = begin; i < end; ++i) {
        const int dbin = kBins - 1 - (int)key_of(lrow[i]);
        bool      in   = false;
        if (dbin < jc) {
            s_a[pa++] = ((unsigned long long)dbin << 32) | (uint32_t)i;
            in        = true;
        }
        else if (dbin == jc) {
            if (pb < nB) {
                if (rec) {
                    vals[nA + pb] = lp_of(dbin);
                    idx[nA + pb]  = i;
                }
                in = true;
            }
            ++pb;
        }
        if (i == sel) {
            s_i[3] = in ? 1 : 0;
            if (rec) {
                out_sel[row] = lp_of(dbin);
            }
        }
    }
    __syncthreads();
    // bitonic sort of the strictly-better list by (descending bin index, id): 1024 slots, unused = ~0
    for (int k = 2; k <= 1024; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int partner = tid ^ j;
            if (partner > tid) {
                const unsigned long long x = s_a[tid], y = s_a[partner];
                const bool up = (tid & k) == 0;
                if ((x > y) == up) {
                    s_a[tid]     = y;
                    s_a[partner] = x;
                }
            }
            __syncthreads();
        }
    }
    if (rec) {
        if (tid < nA) {
            const unsigned long long e = s_a[tid];
            vals[tid] = lp_of((int)(e >> 32));
            idx[tid]  = (int)(uint32_t)e;
        }
        if (tid == 0) {
            out_num[row] = (int)L;
            if (s_i[3] < 0) {            // a drawn id outside the row (cannot happen behind sample_select_kernel)
                out_sel[row] = 0.f;
            }
        }
    }
    if (rec && force_last && n > cap && s_i[3] == 0) {
        __syncthreads();                 // entry cap - 1 was written above by another thread
        if (tid == 0 && sel >= 0 && sel < V) {
            vals[cap - 1] = lp_of(kBins - 1 - (int)key_of(lrow[sel]));
            idx[cap - 1]  = sel;
        }
    }
}

// Philox4x32-10 (Salmon et al., SC'11): counter = (ctr, 0, 0, 0), key = seed; u = top 24 bits / 2^24 in [0, 1)
__host__ __device__ inline float philox_uniform(uint64_t seed, uint32_t ctr)
{
    uint32_t c0 = ctr, c1 = 0, c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

__global__ void philox_uniform_kernel(float* u, const uint64_t* seeds, const int* counters, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) {
        u[b] = philox_uniform(seeds[b], (uint32_t)counters[b]);
    }
}

size_t sample_workspace_bytes(int batch)
{
    return (size_t)batch * kBins * sizeof(uint32_t);
}

int launch_sample_uniform(float* u, const uint64_t* seeds, const int* counters, int batch, hipStream_t st)
{
    if (batch == 0) {
        return 0;
    }
    philox_uniform_kernel<<<(batch + 63) / 64, 64, 0, st>>>(u, seeds, counters, batch);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// workspace: batch * 65536 uint32, ZERO on entry (the kernel leaves it zero again)
// lp != nullptr: the kept candidates' logprobs as well (sample_logprobs_kernel); kept_out must then be given
int launch_sample(int* out_ids, int* kept_out, const half_t* logits, int batch, int V, int ld, const float* temperature,
                  const int* top_k, const float* top_p, const float* min_p, const float* uniform, void* workspace,
                  hipStream_t st, const SampleLogprobs* lp)
{
    TM_REQUIRE(out_ids && logits && workspace, "null pointer");
    TM_REQUIRE(V >= 1 && ld >= V && ld % 8 == 0, "sampling: ld must be a multiple of 8 and >= vocab");
    if (lp) {
        TM_REQUIRE(kept_out && lp->vals && lp->idx && lp->num && lp->sel, "sampling logprobs: null pointer");
        TM_REQUIRE(lp->cap >= 1 && lp->cap <= kMaxLogProb, "sampling logprobs: 1 <= cap <= 1024");
    }
    if (batch == 0) {
        return 0;
    }
    const int chunks = std::max(1, std::min(64, (V + 2047) / 2048));
    sample_hist_kernel<<<dim3(chunks, batch), 256, 0, st>>>((uint32_t*)workspace, logits, V, ld);
    TM_HIP_CHECK(hipGetLastError());
    sample_select_kernel<<<batch, 1024, 0, st>>>(out_ids, (uint32_t*)workspace, logits, V, ld, temperature, top_k, top_p,
                                                 min_p, uniform, kept_out, lp ? 1 : 0);
    TM_HIP_CHECK(hipGetLastError());
    if (lp) {
        sample_logprobs_kernel<<<batch, 1024, 0, st>>>(lp->vals, lp->idx, lp->num, lp->sel, lp->cap, lp->cap == kMaxLogProb,
                                                       (uint32_t*)workspace, logits, V, ld, temperature, kept_out, out_ids,
                                                       lp->step, lp->step_stride, lp->row_stride, lp->row0, lp->max_steps);
        TM_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

float philox_uniform_host(uint64_t seed, uint32_t ctr)
{
    return philox_uniform(seed, ctr);
}

// ------------------------------------------------------------------------------------------------------------------
// Logits processors: repetition penalty -> bad ids -> min-length ban of the end ids (the reference's order,
// generation/logits_processor.cc:66-115; temperature is part of the sampler above).
//   * repetition penalty (RepetitionPenaltyKernel, kernels/sampling_penalty_kernels.cu:137-175): every token id that
//     occurs in the sequence so far (prompt + generated, each id once) gets  l < 0 ? l * p : l / p.  The reference
//     rebuilds a bitmask of the whole history in shared memory every step; here the bitmask of a batch slot is
//     PERSISTENT in HBM ([slot][ceil(vocab / 32)] words), cleared at admission and extended by seen_update_kernel with
//     the tokens each forward consumes -- one atomicOr per new token instead of a pass over the history.
//   * bad ids (BanBadWordsKernel single-token case, kernels/ban_bad_words.cu:51-95): l = -max, ids <= 0 are skipped
//     like in the reference.
//   * min length (batchApplyMinLengthPenalty, sampling_penalty_kernels.cu:198-215): end ids (> 0) are banned while
//     k_len + 1 < min_len, k_len = context length of this forward, min_len = prompt length + min_new_tokens.
// The reference processes fp32 copies of the fp16 logits; here the result is rounded back to fp16 once (the sampler
// and arg-max consume fp16): penalised values differ from the reference's by at most half an fp16 ulp.
// ------------------------------------------------------------------------------------------------------------------
__global__ void seen_update_kernel(uint32_t* __restrict__ seen, int words, const int* __restrict__ ids,
                                   const int* __restrict__ cu_q, int nseq, int n_tokens, int vocab, const int* __restrict__ active)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tokens) {
        return;
    }
    if (active && !cu_q && !active[t]) {
        return;  // decode row of a slot that holds no sequence (parked, or being prefilled by this very forward): its token is stale
    }
    int row = t;
    if (cu_q) {  // packed prefill rows: sequence of token t = last r with cu_q[r] <= t
        int lo = 0, hi = nseq - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cu_q[mid] <= t) {
                lo = mid;
            }
            else {
                hi = mid - 1;
            }
        }
        row = lo;
    }
    const int id = ids[t];
    if (id >= 0 && id < vocab) {
        atomicOr(&seen[(size_t)row * words + (id >> 5)], 1u << (id & 31));
    }
}

constexpr int kLpBlock = 2048;  // logits per workgroup (256 threads x 8)

__global__ __launch_bounds__(256) void logits_process_kernel(half_t* __restrict__ logits, int V, int ld, int vocab_offset,
                                                             const uint32_t* __restrict__ seen, int words,
                                                             const float* __restrict__ rep, const int* __restrict__ ban,
                                                             const int* __restrict__ end, const int* __restrict__ k_len,
                                                             const int* __restrict__ min_len)
{
    __shared__ uint32_t banned[kLpBlock / 32];
    const int row  = blockIdx.y;
    const int base = blockIdx.x * kLpBlock;  // local column of this workgroup's first logit
    const int tid  = threadIdx.x;
    if (tid < kLpBlock / 32) {
        banned[tid] = 0;
    }
    __syncthreads();
    // the (few) banned ids of this row that fall into this workgroup's range
    const bool ban_end = k_len[row] + 1 < min_len[row];
    if (tid < kMaxBadIds + kMaxEndIds) {
        const int id = tid < kMaxBadIds ? ban[row * kMaxBadIds + tid] : (ban_end ? end[row * kMaxEndIds + tid - kMaxBadIds] : -1);
        const int c  = id - vocab_offset - base;
        if (id > 0 && c >= 0 && c < kLpBlock && base + c < V) {
            atomicOr(&banned[c >> 5], 1u << (c & 31));
        }
    }
    __syncthreads();
    const int c0 = base + tid * 8;
    if (c0 >= V) {
        return;
    }
    const float    p    = rep[row];
    const bool     on   = p != 1.f && p > 0.f;
    const int      gid  = vocab_offset + c0;  // multiple of 8: the eight ids share one mask word
    const uint32_t smsk = on ? (seen[(size_t)row * words + (gid >> 5)] >> (gid & 31)) & 0xffu : 0u;
    const uint32_t bmsk = (banned[(tid * 8) >> 5] >> ((tid * 8) & 31)) & 0xffu;
    if (!(smsk | bmsk)) {
        return;
    }
    half_t* lp = logits + (size_t)row * ld + c0;
    if (c0 + 8 <= V) {
        half8_t v = *(const half8_t*)lp;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float x = (float)v[i];
            if (smsk >> i & 1) {
                x = x < 0.f ? x * p : x / p;
            }
            v[i] = (bmsk >> i & 1) ? (half_t)-65504.f : (half_t)x;
        }
        *(half8_t*)lp = v;
    }
    else {
        for (int i = 0; c0 + i < V; ++i) {
            float x = (float)lp[i];
            if (smsk >> i & 1) {
                x = x < 0.f ? x * p : x / p;
            }
            lp[i] = (bmsk >> i & 1) ? (half_t)-65504.f : (half_t)x;
        }
    }
}

int launch_seen_update(uint32_t* seen, int words, const int* ids, const int* cu_q, int nseq, int n_tokens, int vocab,
                       hipStream_t st, const int* active)
{
    if (n_tokens <= 0) {
        return 0;
    }
    seen_update_kernel<<<(n_tokens + 255) / 256, 256, 0, st>>>(seen, words, ids, cu_q, nseq, n_tokens, vocab, active);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_logits_process(half_t* logits, int batch, int V, int ld, int vocab_offset, const uint32_t* seen, int words,
                          const float* rep, const int* ban, const int* end, const int* k_len, const int* min_len,
                          hipStream_t st)
{
    TM_REQUIRE(batch >= 1 && V >= 1 && ld >= V, "logits_process: shape");
    TM_REQUIRE(ld % 8 == 0 && vocab_offset % 8 == 0, "logits_process: row stride and vocabulary offset must be multiples of 8");
    TM_REQUIRE((int64_t)words * 32 >= (int64_t)vocab_offset + V, "logits_process: seen mask narrower than the vocabulary");
    dim3 grid((V + kLpBlock - 1) / kLpBlock, batch);
    logits_process_kernel<<<grid, 256, 0, st>>>(logits, V, ld, vocab_offset, seen, words, rep, ban, end, k_len, min_len);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
