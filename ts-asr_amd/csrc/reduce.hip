// Deterministic partial-sum reductions, immediate or batched.
//
// Every parameter gradient on the path ends in the same shape of work: out[c] (+)= sum_p part[p][c] over the per-workgroup
// partial rows a backward kernel left in its workspace (LayerNorm gamma/beta, biases, conv filters, dropout-add biases) or over
// the split-K fp32 slabs of a weight-gradient GEMM. Launched one by one these are ~310 tiny kernels per training step, each
// paying the ~4.7 us floor of a dependent launch inside the step's hipGraph (1.5 ms of a 27 ms step). While deferral is on
// (gradient arena, between begin_backward and finish_backward) the producers only QUEUE a job; tsasr_reduce_flush then runs all
// of them in ONE launch. Summation order inside a job is fixed (interleaved row slices, combined in index order), so results
// are bit-identical between the immediate and the batched path and from run to run - no float atomics anywhere.
#include <string.h>

#include <vector>

#include "common.h"

struct ReduceJob {
    const float *src;     // partial rows: src[p * pstride + c]
    float *dst;           // dst[c] (+)= sum_p
    long long pstride;
    int nparts, width, accumulate, tile0, wide, pad_;   // wide: 0 = tall job, n = wide job with n 1024-column pieces per tile
};

// tall jobs (many partial rows, few columns): tile = 64 columns (256-byte row pieces), 4 row slices per workgroup, combined through LDS
// wide jobs (few slabs, many columns):        tile = 1024 columns, one thread = 4 consecutive columns, 16-byte slab loads
#define RD_WIDE_TILE 1024
#define RD_TALL_COLS 64
static int wide_iters() {   // 1024-column pieces per wide tile (A/B knob; fewer, larger workgroups amortise the job lookup)
    static const int it = 1;
    return it < 1 ? 1 : it;
}

__device__ __forceinline__ void reduce_tile(const ReduceJob &j, int lt, float (*red)[RD_TALL_COLS + 1]) {
    if (j.wide) {
      for (int it = 0; it < j.wide; ++it) {
        const int c = (lt * j.wide + it) * RD_WIDE_TILE + threadIdx.x * 4;
        if (c >= j.width) return;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (c + 4 <= j.width && ((reinterpret_cast<uintptr_t>(j.src) | (uintptr_t)(j.pstride * 4)) & 15) == 0) {
            const float *base = j.src + c;
            int p = 0;
            for (; p + 4 <= j.nparts; p += 4) {   // four slabs requested together, added in slab order (the sum stays bit-identical)
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4 *>(base + (long long)(p + q) * j.pstride);
#pragma unroll
                for (int q = 0; q < 4; ++q) { s[0] += v[q].x; s[1] += v[q].y; s[2] += v[q].z; s[3] += v[q].w; }
            }
            for (; p < j.nparts; ++p) {
                const float4 v = *reinterpret_cast<const float4 *>(base + (long long)p * j.pstride);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            }
        } else {
            for (int p = 0; p < j.nparts; ++p)
                for (int e = 0; e < 4 && c + e < j.width; ++e) s[e] += j.src[(long long)p * j.pstride + c + e];
        }
        if (c + 4 <= j.width && (reinterpret_cast<uintptr_t>(j.dst) & 15) == 0) {
            float4 *d = reinterpret_cast<float4 *>(j.dst + c);
            float4 o = make_float4(s[0], s[1], s[2], s[3]);
            if (j.accumulate) {
                const float4 a = *d;
                o.x = a.x + s[0]; o.y = a.y + s[1]; o.z = a.z + s[2]; o.w = a.w + s[3];
            }
            *d = o;
        } else {
            for (int e = 0; e < 4 && c + e < j.width; ++e) {
                if (j.accumulate) j.dst[c + e] += s[e];
                else j.dst[c + e] = s[e];
            }
        }
      }
      return;
    }
    const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6, col = lt * RD_TALL_COLS + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < j.width) {
        int n = slice;
        for (; n + 12 < j.nparts; n += 16) {   // four independent loads in flight per lane
            s0 += j.src[(long long)n * j.pstride + col];
            s1 += j.src[(long long)(n + 4) * j.pstride + col];
            s2 += j.src[(long long)(n + 8) * j.pstride + col];
            s3 += j.src[(long long)(n + 12) * j.pstride + col];
        }
        for (; n < j.nparts; n += 4) s0 += j.src[(long long)n * j.pstride + col];
    }
    red[slice][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && col < j.width) {
        const float s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        if (j.accumulate) j.dst[col] += s;
        else j.dst[col] = s;
    }
}

__global__ __launch_bounds__(256) void reduce_one_kernel(ReduceJob job) {
    __shared__ float red[4][RD_TALL_COLS + 1];
    reduce_tile(job, blockIdx.x, red);
}

__global__ __launch_bounds__(256) void reduce_many_kernel(const ReduceJob *__restrict__ jobs, int njobs, int bisect) {
    __shared__ float red[4][RD_TALL_COLS + 1];
    // last job whose first tile <= bid (uniform per workgroup): 64-ary search, one lane per probe - two dependent loads for up
    // to 4096 jobs instead of the ~9 of a bisection (the lookup latency, not the slab traffic, bounded this kernel)
    const int bid = blockIdx.x, lane = threadIdx.x & 63;
    int lo = 0, n = njobs;
    if (bisect) {   // A/B only
        int hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].tile0 <= bid) lo = mid; else hi = mid - 1;
        }
        n = 1;
    }
    while (n > 1) {
        const int step = (n + 63) >> 6, idx = lo + lane * step;
        const bool le = lane * step < n && jobs[idx].tile0 <= bid;   // monotone: a prefix of lanes (lane 0 always) says yes
        const int seg = __popcll(__ballot(le)) - 1;
        lo += seg * step;
        n = min(step, n - seg * step);
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const ReduceJob j = jobs[lo];
    reduce_tile(j, bid - j.tile0, red);
}

static std::vector<ReduceJob> g_jobs;
static std::vector<hipStream_t> g_job_streams;   // the stream each queued job's partial rows are produced on
static int g_defer = 0, g_tiles = 0;

static int job_tiles(const ReduceJob &j) { return j.wide ? cdiv(j.width, RD_WIDE_TILE * j.wide) : cdiv(j.width, RD_TALL_COLS); }

bool tsasr_reduce_deferring() { return g_defer != 0; }

// dst[c] (+)= sum_{p < nparts} src[p * pstride + c], c < width: queued while deferral is on, else launched on `st`
void tsasr_reduce_submit(const float *src, float *dst, long long pstride, int nparts, int width, int accumulate, hipStream_t st) {
    if (!dst || width <= 0 || nparts <= 0) return;
    ReduceJob j{src, dst, pstride, nparts, width, accumulate, g_tiles, nparts < 32 && width >= 4096 ? wide_iters() : 0, 0};
    if (g_defer) {
        g_jobs.push_back(j);
        g_job_streams.push_back(st);
        g_tiles += job_tiles(j);
        return;
    }
    j.tile0 = 0;
    reduce_one_kernel<<<job_tiles(j), 256, 0, st>>>(j);
}

static int reduce_flush_impl(void *table_host, void *table_dev, size_t table_bytes, hipStream_t st, bool only_own) {
    std::vector<ReduceJob> sel, rest;
    std::vector<hipStream_t> rest_streams;
    int tiles = 0, rest_tiles = 0;
    for (size_t i = 0; i < g_jobs.size(); ++i) {
        ReduceJob j = g_jobs[i];
        if (!only_own || g_job_streams[i] == st) { j.tile0 = tiles; tiles += job_tiles(j); sel.push_back(j); }
        else { j.tile0 = rest_tiles; rest_tiles += job_tiles(j); rest.push_back(j); rest_streams.push_back(g_job_streams[i]); }
    }
    if (sel.empty()) return 0;
    const size_t need = sel.size() * sizeof(ReduceJob);
    TSASR_CHECK_ARG(table_host && table_dev && table_bytes >= need, "tsasr_reduce_flush: job table too small (%zu < %zu bytes)", table_bytes, need);
    memcpy(table_host, sel.data(), need);
    // Under stream capture the table is NOT uploaded here: the caller copies table_host -> table_dev itself once the capture has
    // ended (the pointers in it stay valid for the graph's lifetime, each captured graph owns its table pair), so a replay carries
    // no memcpy node (ROCm 7.2 replays of captured memset / memcpy nodes have shown stale payloads).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (cap == hipStreamCaptureStatusNone) {
        hipError_t e = hipMemcpyAsync(table_dev, table_host, need, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            tsasr_set_error("tsasr_reduce_flush: job table upload failed: %s", hipGetErrorString(e));
            return TSASR_E_LAUNCH;
        }
    }
    static const int bisect = 0;
    reduce_many_kernel<<<tiles, 256, 0, st>>>((const ReduceJob *)table_dev, (int)sel.size(), bisect);
    g_jobs.swap(rest);
    g_job_streams.swap(rest_streams);
    g_tiles = rest_tiles;
    TSASR_CHECK_LAUNCH("tsasr_reduce_flush");
    return 0;
}


extern "C" {

/* 1: reductions of partial gradient rows / split-K slabs requested from now on are queued (their partial buffers and outputs
 * must stay alive and untouched until tsasr_reduce_flush); 0: launched immediately (default). Switching off with jobs still
 * queued is an error. Only calls that declare it take part: tsasr_gemm_bf16(accumulate = 2) and the *_bwd kernels' parameter
 * gradient outputs (dgamma, dbeta, dbias, conv-module dparams, fused-GEMM dbias, front-end block dparams, the joint's dW / dbias). */
int tsasr_reduce_defer(int on) {
    TSASR_CHECK_ARG(on || g_jobs.empty(), "tsasr_reduce_defer(0) with %d reductions still queued: call tsasr_reduce_flush first", (int)g_jobs.size());
    g_defer = on ? 1 : 0;
    return 0;
}

int tsasr_reduce_pending(void) { return (int)g_jobs.size(); }

/* Drop every queued reduction without running it and switch deferral off (error paths: a step capture that raised half-way). */
void tsasr_reduce_discard(void) {
    g_jobs.clear();
    g_job_streams.clear();
    g_tiles = 0;
    g_defer = 0;
}

size_t tsasr_reduce_table_bytes(int max_jobs) { return (size_t)max_jobs * sizeof(ReduceJob); }

/* Runs every queued reduction in ONE launch. table_host: PINNED host memory, table_dev: device memory, both `table_bytes` >=
 * tsasr_reduce_table_bytes(tsasr_reduce_pending()); the job table is copied host -> device on `stream` - except while `stream` is
 * being captured: then only table_host is filled and the caller uploads it to table_dev after the capture (both must outlive the graph). */
int tsasr_reduce_flush(void *table_host, void *table_dev, size_t table_bytes, void *stream) {
    return reduce_flush_impl(table_host, table_dev, table_bytes, (hipStream_t)stream, false);
}

/* Same, but only the jobs whose partial rows were produced ON `stream` (they are complete in its order): lets the main stream
 * reduce its share while a forked stream is still running its part of backward; the rest goes with the final tsasr_reduce_flush. */
int tsasr_reduce_flush_stream(void *table_host, void *table_dev, size_t table_bytes, void *stream) {
    return reduce_flush_impl(table_host, table_dev, table_bytes, (hipStream_t)stream, true);
}

}  // extern "C"
