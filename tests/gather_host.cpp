// gather_host.cpp -- test program (tests/test_parity_gpu.py::test_c_abi_hand_off_over_rccl_from_a_cpp_host): a C++ host drives the
// multi-GPU hand-off of the C ABI (include/lsd_hip.h) the way an N-GPU job would, with the one GPU there is: RCCL communicator of
// world size 1, mylsd::set_device / context() from the adapter, lsd_shard_range, lsd_enqueue_batch_device on the rank's shard,
// lsd_gather_lines on the same stream, lsd_gather_unpack on host copies.  Prints the offsets and an FNV-1a hash of the line records.
//   gather_host <raw u8 file with n images of rows x cols> n rows cols
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <vector>

#include "myLSD.h"

#define CK(x) do { if ((x) != 0) { std::fprintf(stderr, "failed: %s\n", #x); return 2; } } while (0)

int main(int argc, char** argv) {
    if (argc < 5) return 1;
    const int n = std::atoi(argv[2]), rows = std::atoi(argv[3]), cols = std::atoi(argv[4]);
    std::vector<uint8_t> host((size_t)n * rows * cols);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(host.data(), 1, host.size(), f) != host.size()) return 1;
    std::fclose(f);

    mylsd::set_device(0);                                   // this "rank"'s GPU
    lsd_ctx* c = mylsd::context();
    ncclUniqueId id;
    ncclComm_t comm;
    CK(ncclGetUniqueId(&id));
    CK(ncclCommInitRank(&comm, 1, id, 0));
    lsd_comm lc;
    CK(lsd_comm_from_rccl(comm, &lc));
    int lo, hi, per;
    size_t words;
    lsd_shard_range(n, lc.world, lc.rank, &lo, &hi);
    CK(lsd_gather_layout(n, lc.world, &per, &words));
    const int n_local = hi - lo, max_lines = 256, cap_rows = n * 256;

    hipStream_t s;
    CK(hipStreamCreate(&s));
    uint8_t* d_maps; lsd_line *d_lines, *d_slabs; int32_t *d_counts, *d_counts_all;
    CK(hipMalloc((void**)&d_maps, host.size()));
    CK(hipMalloc((void**)&d_lines, sizeof(lsd_line) * (size_t)n_local * max_lines));
    CK(hipMalloc((void**)&d_counts, 4 * (size_t)n_local));
    CK(hipMalloc((void**)&d_counts_all, 4 * words));
    CK(hipMalloc((void**)&d_slabs, sizeof(lsd_line) * (size_t)lc.world * cap_rows));
    CK(hipMemcpy(d_maps, host.data() + (size_t)lo * rows * cols, (size_t)n_local * rows * cols, hipMemcpyHostToDevice));
    lsd_params p;
    lsd_default_params(&p);
    CK(lsd_enqueue_batch_device(c, d_maps, n_local, cols, rows, &p, 0, nullptr, d_lines, max_lines, d_counts, s));
    CK(lsd_gather_lines(c, &lc, d_lines, d_counts, n_local, max_lines, n, cap_rows, d_counts_all, d_slabs, s));
    CK(hipStreamSynchronize(s));

    std::vector<int32_t> counts_all(words), offs(n + 1);
    std::vector<lsd_line> slabs((size_t)lc.world * cap_rows), lines((size_t)lc.world * cap_rows);
    CK(hipMemcpy(counts_all.data(), d_counts_all, 4 * words, hipMemcpyDeviceToHost));
    CK(hipMemcpy(slabs.data(), d_slabs, sizeof(lsd_line) * slabs.size(), hipMemcpyDeviceToHost));
    CK(lsd_gather_unpack(counts_all.data(), slabs.data(), n, lc.world, cap_rows, offs.data(), lines.data(), lines.size()));
    unsigned long long h = 1469598103934665603ull;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(lines.data());
    for (size_t i = 0; i < sizeof(lsd_line) * (size_t)offs[n]; i++) { h ^= b[i]; h *= 1099511628211ull; }
    std::printf("world %d rank %d per %d offsets", lc.world, lc.rank, per);
    for (int i = 0; i <= n; i++) std::printf(" %d", offs[i]);
    std::printf(" hash %llu\n", h);
    ncclCommDestroy(comm);
    return 0;
}
