// OSD front end of one frame on one wavefront: reliability sort, column gather, GF(2) elimination, MRB bookkeeping
// (swapped_info / identify_mrb, PB_OSD/pb_testing.py:268-320).  Shared by osd_front_kernel, the fused order-2 kernel
// (ldpc_osd.hip) and the fused PB-OSD head (ldpc_osd_pb.hip).
#pragma once
#include "ldpc_internal.h"
#include "ldpc_wave.h"

namespace ldpc {

struct __attribute__((aligned(16))) FrontLds {
    RankLds rank;            // reliability sort (bucket_ranks, ldpc_wave.h)
    u64 colbuf[64];          // parity columns in primed order
    unsigned mask[4];        // 128-bit membership mask of the MRB indices
    unsigned char pi1[128];  // sorted position -> original bit
    unsigned char rowsrc[64];
    unsigned char perm[128]; // primed position -> original bit
};

struct FrontResult {
    int o1, o2;   // original bit index of primed positions lane and 64 + lane
    u64 Prow;     // row `lane` of P'
    int ns;       // recorded column exchanges (-1: rank-deficient)
};


// sort + column gather + elimination + bookkeeping of one frame (one wavefront); results in registers
// (a1 / a2: the magnitude bits of y[lane] / y[64 + lane])
__device__ __forceinline__ FrontResult front_device_vals(FrontLds &L, unsigned a1, unsigned a2, const u64 *__restrict__ Gcols, int lane)
{
    // ---- reliability sort: rank of each |y| in descending order, ties -> lower index ------
    // sort key = (|y| bits, 127 - index) as one 64-bit integer: "u before v" <=> key_u > key_v (bucket_ranks)
    const float bs = bucket_scale(a1, a2);
    int r1, r2;
    bucket_ranks(L.rank, ((u64)a1 << 32) | (unsigned)(127 - lane), ((u64)a2 << 32) | (unsigned)(63 - lane), bucket_of(a1, bs),
                 bucket_of(a2, bs), lane, r1, r2);
    L.pi1[r1] = (unsigned char)lane;
    L.pi1[r2] = (unsigned char)(lane + 64);
    if (lane < 4) L.mask[lane] = 0;
    wave_fence();
    // ---- G with columns in sorted order, column-major ------------------------------------
    u64 C1 = Gcols[L.pi1[lane]];
    u64 C2 = Gcols[L.pi1[lane + 64]];
    int rho = lane, idx1 = lane, idx2 = lane + 64;
    const int ns = ge_columns(C1, C2, rho, idx1, idx2, lane, nullptr);
    // ---- identify_mrb bookkeeping (pb_testing.py:276-304) --------------------------------
    // (no column exchange -- the 64 most reliable columns were independent: a quarter of the frames -- leaves every index
    //  where the sort put it: the ranks are the lane numbers and the membership mask is not needed)
    int rankM = lane, rankL = lane;
    if (ns != 0) {
        atomicOr(&L.mask[idx1 >> 5], 1u << (idx1 & 31));
        wave_fence();
        const unsigned m[4] = {L.mask[0], L.mask[1], L.mask[2], L.mask[3]};
        rankM = below_mask(m, idx1);         // new MRB position of slot `lane`
        rankL = idx2 - below_mask(m, idx2);  // new parity column of slot `lane`
    }
    L.perm[rankM] = L.pi1[idx1];
    L.perm[64 + rankL] = L.pi1[idx2];
    L.colbuf[rankL] = C2;
    L.rowsrc[rankM] = (unsigned char)rho;          // pivot of MRB slot `lane` is physical row rho
    wave_fence();
    const u64 R = transpose64(L.colbuf[lane], lane);   // lane = physical row, bit = parity column
    FrontResult res;
    res.Prow = shfl64(R, L.rowsrc[lane]);
    res.o1 = L.perm[lane];
    res.o2 = L.perm[64 + lane];
    res.ns = ns;
    wave_fence();
    return res;
}

__device__ __forceinline__ FrontResult front_device(FrontLds &L, const float *__restrict__ y, long long src,
                                                    const u64 *__restrict__ Gcols, int lane)
{
    return front_device_vals(L, __float_as_uint(y[src * 128 + lane]) & 0x7FFFFFFFu, __float_as_uint(y[src * 128 + 64 + lane]) & 0x7FFFFFFFu, Gcols, lane);
}


}  // namespace ldpc
