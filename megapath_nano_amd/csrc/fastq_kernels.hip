// Read quality filter arithmetic (gfx950): per read, the sum of base-call error probabilities and the sum after
// cropping, added in the reference's order so that the doubles are bit-identical (include/mpn_fastq.h).
//
// Data layout in HBM: quality strings concatenated (1 B/base) + CSR offsets; 128-entry probability table in LDS.
// One lane per read: the additions of a read are a dependent chain (that is the point), parallelism comes from the
// reads.  HBM-bound byte scan, 1 B/base in, 16 B/read out.
#include "mpn_common.h"
#include "../../include/mpn_fastq.h"

namespace mpn {

__global__ __launch_bounds__(256) void fastq_qsum_kernel(const uint8_t *__restrict__ qual, const int64_t *__restrict__ off,
                                                         const int32_t *__restrict__ len, int n, int head_crop, int tail_crop,
                                                         int min_len, const double *__restrict__ table, double *__restrict__ total,
                                                         double *__restrict__ cropped, uint8_t *__restrict__ status) {
    __shared__ double tab[128];
    if (threadIdx.x < 128) tab[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *q = qual + off[i];
    const int l = len[i];
    double t = 0;
    bool bad = false;
    for (int k = 0; k < l; ++k) {
        const int c = (int)q[k] - 33;
        if (c < 0 || c > 127) { bad = true; break; }
        t += tab[c];
    }
    double c2 = t;
    const int start = head_crop, end = l - tail_crop;
    if (!bad && end - start >= min_len) {
        for (int k = 0; k < start; ++k) c2 -= tab[(int)q[k] - 33];
        for (int k = end; k < l; ++k) c2 -= tab[(int)q[k] - 33];
    }
    total[i] = bad ? 0.0 : t;
    cropped[i] = bad ? 0.0 : c2;
    status[i] = bad ? 1 : 0;
}

}  // namespace mpn

using namespace mpn;

extern "C" int mpn_fastq_qsum_batch(int32_t n, const uint8_t *qual, const int64_t *off, const int32_t *len, int32_t head_crop,
                                    int32_t tail_crop, int32_t min_len, const double *table, double *total, double *cropped,
                                    uint8_t *status) {
    if (n < 0 || head_crop < 0 || tail_crop < 0 || (n > 0 && (!qual || !off || !len || !table || !total || !cropped || !status))) {
        set_error("mpn_fastq_qsum_batch: bad arguments");
        return -1;
    }
    if (n == 0) return 0;
    int64_t extent = 0;
    for (int i = 0; i < n; ++i) {
        if (len[i] < 0 || off[i] < 0) { set_error("mpn_fastq_qsum_batch: negative offset or length at read %d", i); return -1; }
        extent = std::max<int64_t>(extent, off[i] + len[i]);
    }
    hipStream_t st = 0;
    DevBuf<uint8_t> d_q, d_status;
    DevBuf<int64_t> d_off;
    DevBuf<int32_t> d_len;
    DevBuf<double> d_tab, d_total, d_crop;
    if (d_q.upload(qual, (size_t)extent, st) || d_off.upload(off, n, st) || d_len.upload(len, n, st) || d_tab.upload(table, 128, st) ||
        d_total.alloc(n) || d_crop.alloc(n) || d_status.alloc(n))
        return -1;
    hipLaunchKernelGGL(fastq_qsum_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_q.p, d_off.p, d_len.p, n, head_crop, tail_crop, min_len,
                       d_tab.p, d_total.p, d_crop.p, d_status.p);
    MPN_HIP_CHECK(hipGetLastError());
    if (d_total.download(total, n, st) || d_crop.download(cropped, n, st) || d_status.download(status, n, st)) return -1;
    MPN_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}
