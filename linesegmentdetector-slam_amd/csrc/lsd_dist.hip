// lsd_dist.hip -- the multi-GPU hand-off of liblsdhip.so (include/lsd_hip.h, "multi-GPU"): image shards and the gather of the
// ragged line lists (the reference has no such layer, its callers LSD/main_on_windows.cpp:67-70 and
// main_on_linux.cpp:130-132 run one map on one host).
//
// Images are independent, so a batch is sharded across the GPUs of a node (one process per GPU) with no collective on the data
// path.  The only exchange is the result hand-off: each rank packs its line records into a fixed-capacity slab ON THE DEVICE
// (image-major, compacted), and two regular all-gathers -- the per-image counts, then the slabs -- bring every rank's lists to
// every rank on the caller's stream, with no host synchronisation in between.  The library reaches the collective through a
// three-field communicator (rank, world, all_gather callback): lsd_comm_from_rccl binds it to ncclAllGather of an RCCL
// communicator (RCCL over xGMI on MI355X); tests bind it to a copy loop.
#include <dlfcn.h>
// (only three RCCL entry points are used, through dlsym: their prototypes are restated here so that the library builds -- and loads --
//  without the RCCL headers and library)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;            // ncclSuccess == 0
typedef int ncclDataType_t;          // ncclInt8 == 0
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclInt8 = 0;

#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "lsd_internal.h"

namespace lsdhip {

// cpad[per + 2]: counts of this rank's images clamped to max_lines (zero-padded to `per`), [per] = rows in the slab, [per + 1] = flags:
// bit 0 rows were dropped (more than cap_rows lines, or an image with more than max_lines), bit 1 an image the region stage gave up
// (count < 0: the watchdog).  One workgroup, n_local <= a few thousand
__global__ __launch_bounds__(256) void k_pack_counts(const int32_t* __restrict__ counts, int n_local, int max_lines, int per, int cap_rows,
                                                     int32_t* __restrict__ cpad, int32_t* __restrict__ offs) {
    __shared__ int s_part[256];
    __shared__ int s_over[256];
    const int t = threadIdx.x;
    const int chunk = (n_local + 255) / 256, lo = min(t * chunk, n_local), hi = min(lo + chunk, n_local);
    int sum = 0, over = 0;
    for (int i = lo; i < hi; i++) { const int c = counts[i]; sum += max(0, min(c, max_lines)); over |= (c > max_lines ? 1 : 0) | (c < 0 ? 2 : 0); }
    s_part[t] = sum; s_over[t] = over;
    __syncthreads();
    if (t == 0) {
        int run = 0, ov = 0;
        for (int j = 0; j < 256; j++) { const int v = s_part[j]; s_part[j] = run; run += v; ov |= s_over[j]; }
        cpad[per] = min(run, cap_rows); cpad[per + 1] = ov | (run > cap_rows ? 1 : 0);
    }
    __syncthreads();
    int run = s_part[t];
    for (int i = lo; i < hi; i++) { const int c = max(0, min(counts[i], max_lines)); offs[i] = run; cpad[i] = c; run += c; }
    for (int i = n_local + t; i < per; i += 256) cpad[i] = 0;
}

// image i's records go to rows [offs[i], offs[i] + count) of the slab; rows at or beyond cap_rows are dropped (flagged above)
__global__ __launch_bounds__(256) void k_pack_lines(const lsd_line* __restrict__ lines, const int32_t* __restrict__ cpad, const int32_t* __restrict__ offs,
                                                    int max_lines, int cap_rows, lsd_line* __restrict__ slab) {
    const size_t i = blockIdx.x;
    const int off = offs[i];
    const int keep = max(0, min(cpad[i], cap_rows - off));
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(lines) + i * (size_t)max_lines * 10;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(slab) + (size_t)off * 10;
    for (size_t j = threadIdx.x; j < (size_t)keep * 10; j += 256) dst[j] = src[j];
}

}  // namespace lsdhip

using namespace lsdhip;

extern "C" {

void lsd_shard_range(int n_items, int world, int rank, int* lo, int* hi) {
    if (world <= 0) world = 1;
    const long long n = n_items, w = world, r = rank;
    if (lo) *lo = (int)((r * n + w - 1) / w);
    if (hi) *hi = (int)(((r + 1) * n + w - 1) / w);
}

// Cost-aware deal: perm[] orders the images such that the contiguous shards lsd_shard_range gives the ranks carry about the same cost.
// Longest-processing-time-first: the images by descending cost (ties: ascending index), each to the rank with the least cost so far
// among those whose shard still has room; inside a shard the images keep ascending index.  Deterministic.
int lsd_shard_balanced(const long long* costs, int n_items, int world, int* perm) {
    if (!costs || !perm || n_items <= 0 || world <= 0) return LSD_ERR_INVALID;
    std::vector<int> idx(n_items), room(world), lo(world);
    std::vector<long long> load(world, 0);
    std::vector<std::vector<int>> bin(world);
    for (int r = 0; r < world; r++) { int a, b; lsd_shard_range(n_items, world, r, &a, &b); lo[r] = a; room[r] = b - a; }
    for (int i = 0; i < n_items; i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return costs[a] > costs[b]; });
    for (int i : idx) {
        int best = -1;
        for (int r = 0; r < world; r++)
            if (room[r] > 0 && (best < 0 || load[r] < load[best])) best = r;
        bin[best].push_back(i); load[best] += costs[i] > 0 ? costs[i] : 0; room[best]--;
    }
    for (int r = 0; r < world; r++) {
        std::sort(bin[r].begin(), bin[r].end());
        for (size_t k = 0; k < bin[r].size(); k++) perm[lo[r] + (int)k] = bin[r][k];
    }
    return LSD_OK;
}

int lsd_gather_layout(int n_total, int world, int* per_rank, size_t* counts_words) {
    if (n_total <= 0 || world <= 0) return LSD_ERR_INVALID;
    int per = 0;
    for (int r = 0; r < world; r++) { int lo, hi; lsd_shard_range(n_total, world, r, &lo, &hi); if (hi - lo > per) per = hi - lo; }
    if (per_rank) *per_rank = per;
    if (counts_words) *counts_words = (size_t)world * (size_t)(per + 2);
    return LSD_OK;
}

// ---- RCCL binding: the symbols are looked up at run time in the RCCL the PROCESS has loaded -- the library the communicator handed in
//      belongs to -- so that liblsdhip.so itself loads without RCCL.  A second copy is never loaded: a communicator must not be passed
//      to another instance of the library than the one that made it (LSD_ERR_UNSUPPORTED instead) ----
typedef ncclResult_t (*all_gather_fn)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*comm_int_fn)(const ncclComm_t, int*);
static void* rccl_sym(const char* name) {
    void* p = dlsym(RTLD_DEFAULT, name);                 // linked by the host, or loaded with RTLD_GLOBAL
    if (p) return p;
    // loaded privately (RTLD_LOCAL, e.g. by a Python extension): take a handle to that very copy, do not load one
    static void* h = nullptr;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    return h ? dlsym(h, name) : nullptr;
}
static int rccl_all_gather(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream) {
    static all_gather_fn fn = (all_gather_fn)rccl_sym("ncclAllGather");
    if (!fn) return -1;
    return (int)fn(d_send, d_recv, bytes_per_rank, ncclInt8, (ncclComm_t)user, (hipStream_t)stream);
}

int lsd_comm_from_rccl(void* nccl_comm, lsd_comm* out) {
    if (!nccl_comm || !out) return LSD_ERR_INVALID;
    comm_int_fn cnt = (comm_int_fn)rccl_sym("ncclCommCount"), rnk = (comm_int_fn)rccl_sym("ncclCommUserRank");
    if (!cnt || !rnk || !rccl_sym("ncclAllGather")) return LSD_ERR_UNSUPPORTED;     // no RCCL in this process
    int world = 0, rank = 0;
    if (cnt((ncclComm_t)nccl_comm, &world) != ncclSuccess || rnk((ncclComm_t)nccl_comm, &rank) != ncclSuccess) return LSD_ERR_INVALID;
    out->rank = rank; out->world = world; out->all_gather = rccl_all_gather; out->user = nccl_comm;
    return LSD_OK;
}

int lsd_gather_unpack(const int32_t* counts_all, const lsd_line* slabs_all, int n_total, int world, int cap_rows, int32_t* offsets_out,
                      lsd_line* lines_out, size_t lines_cap) {
    if (!counts_all || !offsets_out || n_total <= 0 || world <= 0 || cap_rows < 0) return LSD_ERR_INVALID;
    int per = 0;
    lsd_gather_layout(n_total, world, &per, nullptr);
    size_t total = 0;
    int flags = 0;
    for (int r = 0; r < world; r++) {
        const int32_t* c = counts_all + (size_t)r * (per + 2);
        int lo, hi;
        lsd_shard_range(n_total, world, r, &lo, &hi);
        flags |= c[per + 1];
        size_t row = 0;
        for (int i = lo; i < hi; i++) {
            offsets_out[i] = (int32_t)total;
            int k = c[i - lo];
            if (row + (size_t)k > (size_t)c[per]) k = (int)((size_t)c[per] > row ? (size_t)c[per] - row : 0);   // rows the slab dropped
            if (lines_out && slabs_all) {
                if (total + (size_t)k > lines_cap) return LSD_ERR_INVALID;
                memcpy(lines_out + total, slabs_all + (size_t)r * cap_rows + row, sizeof(lsd_line) * (size_t)k);
            }
            row += (size_t)k; total += (size_t)k;
        }
    }
    offsets_out[n_total] = (int32_t)total;
    // (either way offsets and lines hold every record that did arrive)
    return (flags & 2) ? LSD_ERR_INTERNAL : (flags & 1) ? LSD_ERR_CAPACITY : LSD_OK;
}

}  // extern "C"

// (lsd_gather_lines needs the context's workspace: it lives in lsd_ctx.hip next to the struct)
namespace lsdhip {
void launch_pack_lines(const lsd_line* lines, const int32_t* counts, int n_local, int max_lines, int per, int cap_rows, int32_t* cpad,
                       int32_t* offs, lsd_line* slab, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_counts, dim3(1), dim3(256), 0, s, counts, n_local, max_lines, per, cap_rows, cpad, offs);
    if (n_local > 0) hipLaunchKernelGGL(k_pack_lines, dim3(n_local), dim3(256), 0, s, lines, cpad, offs, max_lines, cap_rows, slab);
}
}  // namespace lsdhip
