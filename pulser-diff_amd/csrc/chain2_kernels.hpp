// chain2_kernels.hpp — chained two-layout passes with SUB-TILE PIPELINING (included by rydiff.hip after chain_kernels.hpp).
//
// Same algebra as k_chain (finish factor j, start factor j+1; two vectors in, two out).  What changes is the time line
// inside a workgroup.  At N=20, B=1 there are exactly 256 tiles for 256 CUs, so every CU runs ONE workgroup whose phases
// (load all / barrier / compute / store all) cannot overlap with anything: the CU's memory pipe sits idle while it
// computes and runs reads and writes one after the other (measured: 13.4 us of memory time per pass against 9.2 us for
// the same bytes streamed; profiles/r01_bw_probe.txt, tools/bw_probe3.hip).
//
// Here the 2^12 super-tile (same coalesced footprint as before) is split into TWO INDEPENDENT sub-tiles of 2^11
// amplitudes by excluding one index bit from the tile-local flips — a bit that the OTHER layout covers:
//     layout A  super-tile bits [0,12)              excluded bit  eA = lo_B-1   (sub-tiles = alternating 2^eA-blocks)
//     layout B  super-tile bits [0,lo_B) u [12,N)   excluded bit  eB = lo_B-2
// so A-sub u B-sub still covers all N bits (lo_B = 24-N >= 2, i.e. N <= 22).  All loads of both sub-tiles are issued
// up front; sub-tile 0 is finished / stored / started / stored while the loads of sub-tile 1 are still landing, and its
// stores drain while sub-tile 1 computes.  Reads and writes overlap again inside the CU.
#pragma once

struct Chain2Args {
    const double2* u;
    const double2* p;
    double2* v_out;
    double2* q_out;
    const double* utt;   // split diagonal of this layout (indexed by the super-tile index), see k_build_split
    const double* vr;
    const double* coef_fin;
    const double* coef_sta;
    long coef_bstride;
    double fb_r, fb_i, sg_r, sg_i, sb_r, sb_i;
    int lo, hs, hb;      // super-tile layout (as in ChainArgs)
    int sub_bit;         // super-tile index bit that selects the sub-tile (excluded from the local flips)
    uint32_t dim;
    int has_p, has_q, write_v;
    int ga, gd;
    uint32_t fin_mask[kMaxGroups];   // masks over the 11 SUB-TILE-LOCAL bits
    uint32_t sta_mask[kMaxGroups];
    uint32_t dmask[kMaxGroups];
    int dcnt[kMaxGroups];
    // backward mode
    const double2* x_fin;
    const double2* x_sta;
    double* ge_fin;
    double* ge_sta;
    long ge_bstride, ge_rstride;
    double cb_fin_r, cb_fin_i, cb_sta_r, cb_sta_i;
    double* wtot;
    // fused expectation (forward)
    const double* obs;
    double* expect_slot;
    int n_obs;
    long exp_ostride;
};

template <int LGT, bool CPLX, bool BWD>
__global__ __launch_bounds__(1 << LGT) void k_chain2(Chain2Args a) {
    constexpr int LS = 11;                       // sub-tile bits
    constexpr int NT = 1 << LGT, R = 1 << (LS - LGT);
    extern __shared__ __attribute__((aligned(16))) double2 lds[];
    double2* tile0 = lds;
    double2* tile1 = lds + (1 << LS);
    double* red = reinterpret_cast<double*>(lds + (2 << LS));
    const unsigned tid = threadIdx.x;
    const unsigned t = blockIdx.x;
    const size_t boff = size_t(blockIdx.y) * a.dim;
    const unsigned lomask = (1u << a.lo) - 1u;
    const int midlow = a.hs - a.lo;
    const unsigned xbase = ((t & ((1u << midlow) - 1u)) << a.lo) | ((t >> midlow) << (a.hs + a.hb));
    const unsigned sb = unsigned(a.sub_bit);
    const unsigned below = (1u << sb) - 1u;

    // ---- issue every load of both sub-tiles now ---------------------------------------------------------------------
    double2 uu[2][R], pp[2][R], xf[2][R];
    double dg[2][R];
    unsigned xg[2][R];
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // sub-tile 0's loads are issued (and hence retire: vmcnt is in order) before sub-tile 1's
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned j = unsigned(r) * NT + tid;                                   // sub-tile-local index
            const unsigned i = ((j & ~below) << 1) | (unsigned(k) << sb) | (j & below);   // super-tile index
            xg[k][r] = xbase | (i & lomask) | ((i >> a.lo) << a.hs);
            uu[k][r] = a.u[boff + xg[k][r]];
        }
        if (a.has_p) {
#pragma unroll
            for (int r = 0; r < R; ++r) pp[k][r] = a.p[boff + xg[k][r]];
        }
        if (BWD && a.has_p) {
#pragma unroll
            for (int r = 0; r < R; ++r) xf[k][r] = a.x_fin[boff + xg[k][r]];
        }
        if (a.has_q) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned j = unsigned(r) * NT + tid;
#ifndef RYDIFF_ABLATE_DIAG
                dg[k][r] = a.utt[((j & ~below) << 1) | (unsigned(k) << sb) | (j & below)];
#else
                dg[k][r] = 0.0;
#endif
            }
        }
    }
    // interaction diagonal: remote part + cross terms of the super-tile bits (n = 1 - bit)
    double vloc[12];
    double dlane = 0.0;
    if (a.has_q) {
        const double* __restrict__ vrow = a.vr + size_t(t) * 16;
#pragma unroll
        for (int b2 = 0; b2 < 12; ++b2) vloc[b2] = vrow[b2];
        dlane = vrow[12];
    }
    const double* __restrict__ cff = a.coef_fin + blockIdx.y * a.coef_bstride;
    const double* __restrict__ cfs = a.coef_sta + blockIdx.y * a.coef_bstride;
    double* ge_fin = nullptr;
    double* ge_sta = nullptr;
    if (BWD) {
        const long goff = blockIdx.y * a.ge_bstride + (blockIdx.x % kGradReplicas) * a.ge_rstride;
        ge_fin = a.ge_fin + goff;
        ge_sta = a.ge_sta + goff;
    }

#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double2* tile = k ? tile1 : tile0;
        double2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) tile[unsigned(r) * NT + tid] = uu[k][r];
        __syncthreads();
        // ---- finish factor j on this sub-tile ---------------------------------------------------------------------
        if (a.has_p) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = pp[k][r];
            for (int g = 0; g < a.ga; ++g) {
                const uint32_t mask = a.fin_mask[g];
                if (!mask) continue;
                double2 ts[R], ds[R];
#ifndef RYDIFF_ABLATE_COMPUTE
                partner_sums<LS, LGT, CPLX || BWD>(tile, uu[k], mask, tid, ts, ds);
#else
                for (int r = 0; r < R; ++r) { ts[r] = uu[k][r]; ds[r] = uu[k][r]; }
#endif
                const double cr = cff[g], ci = cff[a.ga + g];
                const double k1r = a.fb_r * cr, k1i = a.fb_i * cr, k2r = -a.fb_i * ci, k2i = a.fb_r * ci;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r].x += k1r * ts[r].x - k1i * ts[r].y;
                    acc[r].y += k1r * ts[r].y + k1i * ts[r].x;
                    if (CPLX) {
                        acc[r].x += k2r * ds[r].x - k2i * ds[r].y;
                        acc[r].y += k2r * ds[r].y + k2i * ds[r].x;
                    }
                }
                if (BWD) {
                    double z1r = 0.0, z1i = 0.0, z2r = 0.0, z2i = 0.0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        z1r += ts[r].x * xf[k][r].x + ts[r].y * xf[k][r].y;
                        z1i += ts[r].x * xf[k][r].y - ts[r].y * xf[k][r].x;
                        z2r += ds[r].x * xf[k][r].x + ds[r].y * xf[k][r].y;
                        z2i += ds[r].x * xf[k][r].y - ds[r].y * xf[k][r].x;
                    }
                    wg_atomic_add<NT>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, ge_fin + g, red);
                    wg_atomic_add<NT>(a.cb_fin_r * z2i + a.cb_fin_i * z2r, ge_fin + a.ga + g, red);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = uu[k][r];
        }
        double2 xs[R];
        if (BWD && a.has_q) {
#pragma unroll
            for (int r = 0; r < R; ++r) xs[r] = a.x_sta[boff + xg[k][r]];
        }
        if (a.write_v) {
#pragma unroll
            for (int r = 0; r < R; ++r) stream_store(a.v_out + boff + xg[k][r], acc[r]);
        }
        if (!BWD && a.obs) {
            for (int o = 0; o < a.n_obs; ++o) {
                double e = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    e += a.obs[size_t(o) * a.dim + xg[k][r]] * (acc[r].x * acc[r].x + acc[r].y * acc[r].y);
                wg_atomic_add<NT>(e, a.expect_slot + o * a.exp_ostride + blockIdx.y, red);
            }
        }
        if (!a.has_q) continue;

        __syncthreads();  // partner reads of u (this sub-tile) are done
#pragma unroll
        for (int r = 0; r < R; ++r) tile[unsigned(r) * NT + tid] = acc[r];
        __syncthreads();
        // ---- start factor j+1 on this sub-tile -------------------------------------------------------------------
        double2 q[R];
        double rr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned j = unsigned(r) * NT + tid;
            const unsigned i = ((j & ~below) << 1) | (unsigned(k) << sb) | (j & below);
            double d = dg[k][r] + dlane;
#pragma unroll
            for (int b2 = 0; b2 < 12; ++b2)
                if (!(i >> b2 & 1u)) d += vloc[b2];
            for (int g = 0; g < a.gd; ++g) d += cfs[2 * a.ga + g] * double(a.dcnt[g] - __popc(xg[k][r] & a.dmask[g]));
            const double dr = a.sg_r + a.sb_r * d, di = a.sg_i + a.sb_i * d;
            q[r].x = dr * acc[r].x - di * acc[r].y;
            q[r].y = dr * acc[r].y + di * acc[r].x;
            if (BWD) {
                const double pr = a.cb_sta_r * acc[r].x + a.cb_sta_i * acc[r].y, pi = a.cb_sta_i * acc[r].x - a.cb_sta_r * acc[r].y;
                rr[r] = pr * xs[r].x - pi * xs[r].y;
                if (a.wtot) unsafeAtomicAdd(a.wtot + xg[k][r], rr[r]);
            }
        }
        if (BWD) {
            for (int g = 0; g < a.gd; ++g) {
                double sgd = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) sgd += rr[r] * double(a.dcnt[g] - __popc(xg[k][r] & a.dmask[g]));
                wg_atomic_add<NT>(sgd, ge_sta + 2 * a.ga + g, red);
            }
        }
        for (int g = 0; g < a.ga; ++g) {
            const uint32_t mask = a.sta_mask[g];
            if (!mask) continue;
            double2 ts[R], ds[R];
#ifndef RYDIFF_ABLATE_COMPUTE
            partner_sums<LS, LGT, CPLX || BWD>(tile, acc, mask, tid, ts, ds);
#else
            for (int r = 0; r < R; ++r) { ts[r] = acc[r]; ds[r] = acc[r]; }
#endif
            const double cr = cfs[g], ci = cfs[a.ga + g];
            const double k1r = a.sb_r * cr, k1i = a.sb_i * cr, k2r = -a.sb_i * ci, k2i = a.sb_r * ci;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                q[r].x += k1r * ts[r].x - k1i * ts[r].y;
                q[r].y += k1r * ts[r].y + k1i * ts[r].x;
                if (CPLX) {
                    q[r].x += k2r * ds[r].x - k2i * ds[r].y;
                    q[r].y += k2r * ds[r].y + k2i * ds[r].x;
                }
            }
            if (BWD) {
                double z1r = 0.0, z1i = 0.0, z2r = 0.0, z2i = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    z1r += ts[r].x * xs[r].x + ts[r].y * xs[r].y;
                    z1i += ts[r].x * xs[r].y - ts[r].y * xs[r].x;
                    z2r += ds[r].x * xs[r].x + ds[r].y * xs[r].y;
                    z2i += ds[r].x * xs[r].y - ds[r].y * xs[r].x;
                }
                wg_atomic_add<NT>(a.cb_sta_r * z1r - a.cb_sta_i * z1i, ge_sta + g, red);
                wg_atomic_add<NT>(a.cb_sta_r * z2i + a.cb_sta_i * z2r, ge_sta + a.ga + g, red);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) stream_store(a.q_out + boff + xg[k][r], q[r]);
    }
}
