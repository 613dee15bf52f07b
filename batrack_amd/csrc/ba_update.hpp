// ba_update.hpp — the part of a step's last kernel that is not per tile, shared by k_update / k_tile (ba_kernels.hip) and
// k_etile (ba_etile.hip): the pose retraction Exp(dX) * G (groups.py:153-156) and the copy + clamp of the patch buffer
// (ba.py:333).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_kernels.hpp"

namespace bt {

__device__ inline void retract_pose(const float *pin, const float *xi, float *pout) {
    // poses' = Exp(xi) * G  (groups.py:153-156; so3.h:153-190; se3.h:134-142), in double
    const double tau[3] = {xi[0], xi[1], xi[2]}, phi[3] = {xi[3], xi[4], xi[5]};
    const double th2 = phi[0]*phi[0] + phi[1]*phi[1] + phi[2]*phi[2], th = sqrt(th2);
    double imag, real, c1, c2;
    if (th < 1e-6) {
        const double th4 = th2 * th2;
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
        real = 1.0 - th2 / 8.0 + th4 / 384.0;
        c1 = 0.5 - th2 / 24.0;
        c2 = 1.0 / 6.0 - th2 / 120.0;
    } else {
        imag = sin(0.5 * th) / th;
        real = cos(0.5 * th);
        c1 = (1.0 - cos(th)) / th2;
        c2 = (th - sin(th)) / (th2 * th);
    }
    double qe[4] = {imag * phi[0], imag * phi[1], imag * phi[2], real};
    double nq = 1.0 / sqrt(qe[0]*qe[0] + qe[1]*qe[1] + qe[2]*qe[2] + qe[3]*qe[3]);
    for (int c = 0; c < 4; ++c) qe[c] *= nq;
    const double pxt[3] = {phi[1]*tau[2] - phi[2]*tau[1], phi[2]*tau[0] - phi[0]*tau[2], phi[0]*tau[1] - phi[1]*tau[0]};
    const double ppt[3] = {phi[1]*pxt[2] - phi[2]*pxt[1], phi[2]*pxt[0] - phi[0]*pxt[2], phi[0]*pxt[1] - phi[1]*pxt[0]};
    double te[3];
    for (int c = 0; c < 3; ++c) te[c] = tau[c] + c1 * pxt[c] + c2 * ppt[c];
    double q[4] = {pin[3], pin[4], pin[5], pin[6]};
    nq = 1.0 / sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    for (int c = 0; c < 4; ++c) q[c] *= nq;
    const double t[3] = {pin[0], pin[1], pin[2]};
    double qo[4] = { qe[3]*q[0] + qe[0]*q[3] + qe[1]*q[2] - qe[2]*q[1],
                     qe[3]*q[1] - qe[0]*q[2] + qe[1]*q[3] + qe[2]*q[0],
                     qe[3]*q[2] + qe[0]*q[1] - qe[1]*q[0] + qe[2]*q[3],
                     qe[3]*q[3] - qe[0]*q[0] - qe[1]*q[1] - qe[2]*q[2] };
    nq = 1.0 / sqrt(qo[0]*qo[0] + qo[1]*qo[1] + qo[2]*qo[2] + qo[3]*qo[3]);
    double ux = qe[1]*t[2] - qe[2]*t[1], uy = qe[2]*t[0] - qe[0]*t[2], uz = qe[0]*t[1] - qe[1]*t[0];
    ux += ux; uy += uy; uz += uz;
    pout[0] = (float)(te[0] + t[0] + qe[3]*ux + (qe[1]*uz - qe[2]*uy));
    pout[1] = (float)(te[1] + t[1] + qe[3]*uy + (qe[2]*ux - qe[0]*uz));
    pout[2] = (float)(te[2] + t[2] + qe[3]*uz + (qe[0]*uy - qe[1]*ux));
    for (int c = 0; c < 4; ++c) pout[3 + c] = (float)(qo[c] * nq);
}

// patch `gid` < p_tot of the buffer is copied and clamped (ba.py:333; TRACKS_ELSEWHERE: patches that carry a track are
// written by the tile blocks and skipped here, else — the unfused structure-only update — their dZ = Q w' is applied here,
// ba.py:316-317), then one thread per buffer pose: Exp(dX) * G in double (groups.py:153-156) or, structure-only, a plain copy.
template <bool SO, bool TRACKS_ELSEWHERE>
__device__ __forceinline__ void update_rest(const PlanDev &pd, const StepArgs &a, int gid, int do_poses) {
    if (gid < pd.p_tot) {
        // track of this patch, or -1: bitmap + rank (most of the buffer's patches are not in the window)
        const unsigned aw = pd.act_bits[gid >> 5], ab = (unsigned)gid & 31u;
        const bool has = (aw >> ab) & 1u;
        if (TRACKS_ELSEWHERE && has) return;                            // written by its tile's block
        const float x = a.patches[3*gid], y = a.patches[3*gid + 1], d = a.patches[3*gid + 2];
        float dd = d;                                                   // ba.py:333 (whole buffer)
        if (SO && has) {                                                // ba.py:316-317
            const int trk = pd.act_rank[gid >> 5] + __popc(aw & ((1u << ab) - 1u));
            if (a.prec) { const double2 qw = reinterpret_cast<const double2 *>(a.qw)[trk]; dd = (float)((double)d + qw.x * qw.y); }
            else { const float2 qw = a.qw[trk]; dd = d + qw.x * qw.y; }
        }
        dd = dd < 1e-3f ? 1e-3f : dd;
        dd = dd > 10.0f ? 10.0f : dd;
        a.patches_out[3*gid] = x; a.patches_out[3*gid + 1] = y; a.patches_out[3*gid + 2] = dd;
    } else if (do_poses && gid < pd.p_tot + pd.n_buf) {
        const int p = gid - pd.p_tot;
        if (SO) {
            for (int c = 0; c < 7; ++c) a.poses_out[7*p + c] = a.poses[7*p + c];
        } else {
            float xi[6] = {0, 0, 0, 0, 0, 0};
            if (p >= pd.fixedp && p < pd.fixedp + pd.n)
                for (int c = 0; c < 6; ++c) xi[c] = a.dx[6 * (p - pd.fixedp) + c];
            retract_pose(a.poses + 7*p, xi, a.poses_out + 7*p);
        }
    }
}

}  // namespace bt
