/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Included twice by ba_oracle.c with
 * REAL = double / float and SUF = _f64 / _f32.
 *
 * CPU restatement of the reference's bundle-adjustment step
 *   BA_rgbd_droid                 /root/reference/main/backend/ba.py:217-339
 *   transform(jacobian=True)      /root/reference/main/backend/projective_ops.py:54-100
 *   SE3 inv / mul / act4 / adjT / Exp
 *        /root/reference/main/backend/lietorch/include/se3.h:36-67,134-142
 *        /root/reference/main/backend/lietorch/include/so3.h:31-65,153-190
 * written edge-major in straight C (no torch, no BLAS).  Pinned against the
 * golden vectors in tests/golden/ (tests/test_oracle_golden.py).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

typedef struct { REAL t[3]; REAL q[4]; } FN(se3);

static void FN(q_unit)(REAL *q) {            /* so3.h:35-37: normalise on load */
    REAL n = (REAL)sqrt((double)(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]));
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

static void FN(q_rot)(const REAL *q, const REAL *p, REAL *o) {   /* so3.h:55-60 */
    REAL ux = q[1]*p[2] - q[2]*p[1], uy = q[2]*p[0] - q[0]*p[2], uz = q[0]*p[1] - q[1]*p[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = p[0] + q[3]*ux + (q[1]*uz - q[2]*uy);
    o[1] = p[1] + q[3]*uy + (q[2]*ux - q[0]*uz);
    o[2] = p[2] + q[3]*uz + (q[0]*uy - q[1]*ux);
}

static void FN(q_mul)(const REAL *a, const REAL *b, REAL *o) {   /* Hamilton, (x,y,z,w) */
    o[0] = a[3]*b[0] + a[0]*b[3] + a[1]*b[2] - a[2]*b[1];
    o[1] = a[3]*b[1] - a[0]*b[2] + a[1]*b[3] + a[2]*b[0];
    o[2] = a[3]*b[2] + a[0]*b[1] - a[1]*b[0] + a[2]*b[3];
    o[3] = a[3]*b[3] - a[0]*b[0] - a[1]*b[1] - a[2]*b[2];
}

static FN(se3) FN(se3_load)(const REAL *d) {
    FN(se3) g; g.t[0] = d[0]; g.t[1] = d[1]; g.t[2] = d[2];
    g.q[0] = d[3]; g.q[1] = d[4]; g.q[2] = d[5]; g.q[3] = d[6];
    FN(q_unit)(g.q); return g;
}

static FN(se3) FN(se3_inv)(FN(se3) g) {                            /* se3.h:36-38 */
    FN(se3) r; REAL tt[3];
    r.q[0] = -g.q[0]; r.q[1] = -g.q[1]; r.q[2] = -g.q[2]; r.q[3] = g.q[3];
    FN(q_unit)(r.q);
    FN(q_rot)(r.q, g.t, tt);
    r.t[0] = -tt[0]; r.t[1] = -tt[1]; r.t[2] = -tt[2];
    return r;
}

static FN(se3) FN(se3_mul)(FN(se3) a, FN(se3) b) {                 /* se3.h:45-47 */
    FN(se3) r; REAL tt[3];
    FN(q_mul)(a.q, b.q, r.q); FN(q_unit)(r.q);
    FN(q_rot)(a.q, b.t, tt);
    r.t[0] = a.t[0] + tt[0]; r.t[1] = a.t[1] + tt[1]; r.t[2] = a.t[2] + tt[2];
    return r;
}

static void FN(q_matrix)(const REAL *q, REAL R[3][3]) {
    REAL x = q[0], y = q[1], z = q[2], w = q[3];
    R[0][0] = 1 - 2*(y*y + z*z); R[0][1] = 2*(x*y - z*w);     R[0][2] = 2*(x*z + y*w);
    R[1][0] = 2*(x*y + z*w);     R[1][1] = 1 - 2*(x*x + z*z); R[1][2] = 2*(y*z - x*w);
    R[2][0] = 2*(x*z - y*w);     R[2][1] = 2*(y*z + x*w);     R[2][2] = 1 - 2*(x*x + y*y);
}

/* out = Ad(g)^T a, Ad = [[R, [t]x R],[0, R]]                        se3.h:58-67,84-86 */
static void FN(se3_adjT)(FN(se3) g, const REAL *a, REAL *o) {
    REAL R[3][3]; FN(q_matrix)(g.q, R);
    REAL c[3] = { a[1]*g.t[2] - a[2]*g.t[1] + a[3],
                  a[2]*g.t[0] - a[0]*g.t[2] + a[4],
                  a[0]*g.t[1] - a[1]*g.t[0] + a[5] };              /* a_tau x t + a_phi */
    for (int m = 0; m < 3; ++m) {
        o[m]     = R[0][m]*a[0] + R[1][m]*a[1] + R[2][m]*a[2];
        o[3 + m] = R[0][m]*c[0] + R[1][m]*c[1] + R[2][m]*c[2];
    }
}

static FN(se3) FN(se3_exp)(const REAL *xi) {              /* so3.h:153-190, se3.h:134-142 */
    const REAL *tau = xi, *phi = xi + 3;
    REAL th2 = phi[0]*phi[0] + phi[1]*phi[1] + phi[2]*phi[2];
    REAL th = (REAL)sqrt((double)th2), imag, real, c1, c2;
    if (th < 1e-6) {
        REAL th4 = th2 * th2;
        imag = (REAL)0.5 - (REAL)(1.0/48.0)*th2 + (REAL)(1.0/3840.0)*th4;
        real = (REAL)1 - (REAL)(1.0/8.0)*th2 + (REAL)(1.0/384.0)*th4;
        c1 = (REAL)0.5 - (REAL)(1.0/24.0)*th2;
        c2 = (REAL)(1.0/6.0) - (REAL)(1.0/120.0)*th2;
    } else {
        imag = (REAL)sin(0.5 * (double)th) / th;
        real = (REAL)cos(0.5 * (double)th);
        c1 = ((REAL)1 - (REAL)cos((double)th)) / th2;
        c2 = (th - (REAL)sin((double)th)) / (th2 * th);
    }
    FN(se3) g;
    g.q[0] = imag*phi[0]; g.q[1] = imag*phi[1]; g.q[2] = imag*phi[2]; g.q[3] = real;
    FN(q_unit)(g.q);
    REAL a[3] = { phi[1]*tau[2] - phi[2]*tau[1], phi[2]*tau[0] - phi[0]*tau[2], phi[0]*tau[1] - phi[1]*tau[0] };
    REAL b[3] = { phi[1]*a[2] - phi[2]*a[1], phi[2]*a[0] - phi[0]*a[2], phi[0]*a[1] - phi[1]*a[0] };
    for (int m = 0; m < 3; ++m) g.t[m] = tau[m] + c1*a[m] + c2*b[m];
    return g;
}

/* residual weight kernels                                            ba.py:81-100 */
static REAL FN(rho)(REAL r, int loss) {
    REAL s = r * r;
    if (loss == 1) return s > 1 ? (REAL)1 / (REAL)sqrt((double)s) : (REAL)1;
    if (loss == 2) return (REAL)1 / ((REAL)1 + s);
    return (REAL)1;
}

/* One edge: reprojection, Jacobians, masks.  projective_ops.py:54-100, ba.py:228-251 */
typedef struct { REAL uv[2], valid, Ji[2][6], Jj[2][6], Jz[2], r[2], W[2]; } FN(edge_out);

static void FN(edge_eval)(const REAL *poses, const REAL *patches, const REAL *intr,
                          const REAL *target, const REAL *weight, int64_t i, int64_t j, int64_t k,
                          const REAL *bounds, int loss, FN(edge_out) *o) {
    FN(se3) Gi = FN(se3_load)(poses + 7*i), Gj = FN(se3_load)(poses + 7*j);
    FN(se3) Gij = FN(se3_mul)(Gj, FN(se3_inv)(Gi));
    const REAL *Ki = intr + 4*i, *Kj = intr + 4*j;
    REAL x = patches[3*k], y = patches[3*k + 1], d = patches[3*k + 2];
    REAL X0[3] = { (x - Ki[2]) / Ki[0], (y - Ki[3]) / Ki[1], (REAL)1 };   /* iproj :19-29 */
    REAL P[3]; FN(q_rot)(Gij.q, X0, P);
    REAL X = P[0] + Gij.t[0]*d, Y = P[1] + Gij.t[1]*d, Z = P[2] + Gij.t[2]*d, H = d;
    REAL zc = Z < (REAL)1e-2 ? (REAL)1e-2 : Z;                            /* proj :43-45 */
    REAL dz = (REAL)1 / zc;
    o->uv[0] = Kj[0]*(dz*X) + Kj[2];
    o->uv[1] = Kj[1]*(dz*Y) + Kj[3];
    REAL dj = (REAL)fabs((double)Z) > (REAL)0.2 ? (REAL)1 / Z : (REAL)0;   /* :80-81 */
    REAL Jp[2][3] = { { Kj[0]*dj, 0, -Kj[0]*X*dj*dj }, { 0, Kj[1]*dj, -Kj[1]*Y*dj*dj } };
    REAL Ja[3][6] = { { H, 0, 0, 0, Z, -Y }, { 0, H, 0, -Z, 0, X }, { 0, 0, H, Y, -X, 0 } };
    for (int c = 0; c < 2; ++c) {
        for (int m = 0; m < 6; ++m)
            o->Jj[c][m] = Jp[c][0]*Ja[0][m] + Jp[c][1]*Ja[1][m] + Jp[c][2]*Ja[2][m];
        REAL tmp[6]; FN(se3_adjT)(Gij, o->Jj[c], tmp);                     /* :96 */
        for (int m = 0; m < 6; ++m) o->Ji[c][m] = -tmp[m];
        o->Jz[c] = Jp[c][0]*Gij.t[0] + Jp[c][1]*Gij.t[1] + Jp[c][2]*Gij.t[2];   /* :98 */
    }
    REAL r0 = target[0] - o->uv[0], r1 = target[1] - o->uv[1];
    REAL v = Z > (REAL)0.2 ? (REAL)1 : (REAL)0;                            /* :100 */
    v *= ((REAL)sqrt((double)(r0*r0 + r1*r1)) < (REAL)250) ? (REAL)1 : (REAL)0;   /* ba.py:233 */
    v *= (o->uv[0] > bounds[0] && o->uv[1] > bounds[1] &&
          o->uv[0] < bounds[2] && o->uv[1] < bounds[3]) ? (REAL)1 : (REAL)0;
    o->valid = v;
    o->W[0] = v * (weight[0] * FN(rho)(r0, loss));                          /* ba.py:247-251 */
    o->W[1] = v * (weight[1] * FN(rho)(r1, loss));
    o->r[0] = v * r0; o->r[1] = v * r1;
}

/* dense lower Cholesky + solve; returns nonzero if a pivot is not positive (ba.py:9-13) */
static int FN(chol_solve)(REAL *A, REAL *b, int n) {
    for (int c = 0; c < n; ++c) {
        REAL s = A[c*n + c];
        for (int k = 0; k < c; ++k) s -= A[c*n + k]*A[c*n + k];
        if (!(s > 0)) return 1;
        REAL l = (REAL)sqrt((double)s); A[c*n + c] = l;
        for (int r = c + 1; r < n; ++r) {
            REAL t = A[r*n + c];
            for (int k = 0; k < c; ++k) t -= A[r*n + k]*A[c*n + k];
            A[r*n + c] = t / l;
        }
    }
    for (int r = 0; r < n; ++r) { REAL t = b[r]; for (int k = 0; k < r; ++k) t -= A[r*n + k]*b[k]; b[r] = t / A[r*n + r]; }
    for (int r = n - 1; r >= 0; --r) { REAL t = b[r]; for (int k = r + 1; k < n; ++k) t -= A[k*n + r]*b[k]; b[r] = t / A[r*n + r]; }
    return 0;
}

/* Optional per-edge dump for tests (any pointer may be NULL). */
int FN(oracle_edges)(const REAL *poses, const REAL *patches, const REAL *intr,
                     const REAL *targets, int64_t tstride, const REAL *weights,
                     const int64_t *ii, const int64_t *jj, const int64_t *kk, int64_t E,
                     const REAL *bounds, int loss,
                     REAL *coords, REAL *valid, REAL *Ji, REAL *Jj, REAL *Jz, REAL *r, REAL *W) {
    for (int64_t e = 0; e < E; ++e) {
        FN(edge_out) o;
        FN(edge_eval)(poses, patches, intr, targets + tstride*e, weights + 2*e, ii[e], jj[e], kk[e], bounds, loss, &o);
        if (coords) { coords[2*e] = o.uv[0]; coords[2*e + 1] = o.uv[1]; }
        if (valid) valid[e] = o.valid;
        for (int c = 0; c < 2; ++c) {
            for (int m = 0; m < 6; ++m) {
                if (Ji) Ji[12*e + 6*c + m] = o.Ji[c][m];
                if (Jj) Jj[12*e + 6*c + m] = o.Jj[c][m];
            }
            if (Jz) Jz[2*e + c] = o.Jz[c];
            if (r) r[2*e + c] = o.r[c];
            if (W) W[2*e + c] = o.W[c];
        }
    }
    return 0;
}

/* One BA_rgbd_droid call.  Returns 0, or 1 when the Cholesky failed (dX = 0, as
 * the reference), or <0 on bad arguments.  S_out [6n*6n], y_out [6n], dX_out [6n]
 * are optional (undamped S, as handed to block_solve at ba.py:323).
 * loss: 0 trivial, 1 huber, 2 cauchy. */
int FN(oracle_ba_step)(const REAL *poses, const REAL *patches, const REAL *mono, const REAL *intr,
                       const REAL *targets, int64_t tstride, const REAL *weights,
                       const int64_t *ii, const int64_t *jj, const int64_t *kk,
                       int64_t E, int64_t N_buf, int64_t P_tot, const REAL *bounds,
                       REAL lmbda, REAL ep, REAL alpha, int64_t fixedp, int structure_only, int loss,
                       REAL *poses_out, REAL *patches_out, REAL *S_out, REAL *y_out, REAL *dX_out) {
    int64_t n_all = 0;
    for (int64_t e = 0; e < E; ++e) {                                   /* ba.py:219 */
        if (ii[e] + 1 > n_all) n_all = ii[e] + 1;
        if (jj[e] + 1 > n_all) n_all = jj[e] + 1;
        if (kk[e] < 0 || kk[e] >= P_tot || ii[e] < 0 || jj[e] < 0) return -1;
    }
    if (n_all > N_buf) return -1;
    int64_t n = n_all - fixedp; if (n < 0) n = 0;
    int64_t D = 6 * n;

    /* unique(kk), ascending                                            ba.py:276 */
    int64_t *slot = (int64_t *)malloc(sizeof(int64_t) * (size_t)P_tot);
    for (int64_t p = 0; p < P_tot; ++p) slot[p] = -1;
    for (int64_t e = 0; e < E; ++e) slot[kk[e]] = 0;
    int64_t m = 0;
    for (int64_t p = 0; p < P_tot; ++p) if (slot[p] == 0) slot[p] = m++;
    int64_t *kx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    for (int64_t p = 0; p < P_tot; ++p) if (slot[p] >= 0) kx[slot[p]] = p;

    REAL *B = (REAL *)calloc((size_t)(D * D + 1), sizeof(REAL));
    REAL *g = (REAL *)calloc((size_t)(D + 1), sizeof(REAL));
    REAL *C = (REAL *)calloc((size_t)(m + 1), sizeof(REAL));
    REAL *w = (REAL *)calloc((size_t)(m + 1), sizeof(REAL));
    REAL *Ei = (REAL *)malloc(sizeof(REAL) * (size_t)(6 * E + 1));
    REAL *Ej = (REAL *)malloc(sizeof(REAL) * (size_t)(6 * E + 1));

    for (int64_t e = 0; e < E; ++e) {                                   /* ba.py:253-292 */
        FN(edge_out) o;
        FN(edge_eval)(poses, patches, intr, targets + tstride*e, weights + 2*e, ii[e], jj[e], kk[e], bounds, loss, &o);
        int64_t a = ii[e] - fixedp, b = jj[e] - fixedp, k = slot[kk[e]];
        for (int p = 0; p < 6; ++p) {
            REAL wi0 = o.W[0]*o.Ji[0][p], wi1 = o.W[1]*o.Ji[1][p];
            REAL wj0 = o.W[0]*o.Jj[0][p], wj1 = o.W[1]*o.Jj[1][p];
            for (int q = 0; q < 6; ++q) {
                if (a >= 0)           B[(6*a + p)*D + 6*a + q] += wi0*o.Ji[0][q] + wi1*o.Ji[1][q];
                if (a >= 0 && b >= 0) B[(6*a + p)*D + 6*b + q] += wi0*o.Jj[0][q] + wi1*o.Jj[1][q];
                if (a >= 0 && b >= 0) B[(6*b + p)*D + 6*a + q] += wj0*o.Ji[0][q] + wj1*o.Ji[1][q];
                if (b >= 0)           B[(6*b + p)*D + 6*b + q] += wj0*o.Jj[0][q] + wj1*o.Jj[1][q];
            }
            if (a >= 0) g[6*a + p] += wi0*o.r[0] + wi1*o.r[1];
            if (b >= 0) g[6*b + p] += wj0*o.r[0] + wj1*o.r[1];
            Ei[6*e + p] = wi0*o.Jz[0] + wi1*o.Jz[1];
            Ej[6*e + p] = wj0*o.Jz[0] + wj1*o.Jz[1];
        }
        C[k] += o.W[0]*o.Jz[0]*o.Jz[0] + o.W[1]*o.Jz[1]*o.Jz[1];
        w[k] += o.W[0]*o.Jz[0]*o.r[0] + o.W[1]*o.Jz[1]*o.r[1];
    }

    /* depth prior, Q                                                    ba.py:296-311 */
    REAL *Q = (REAL *)malloc(sizeof(REAL) * (size_t)(m + 1));
    for (int64_t k = 0; k < m; ++k) {
        REAL pm = mono[kx[k]] > (REAL)1e-2 ? (REAL)1 : (REAL)0;
        REAL Ca = C[k] + pm*alpha; Ca = Ca + lmbda;
        w[k] = w[k] - pm*alpha*(patches[3*kx[k] + 2] - mono[kx[k]]);
        Q[k] = (REAL)1 / Ca;
    }

    /* group edges by track, build the track's camera-space E, Schur      ba.py:314-323 */
    int64_t *off = (int64_t *)calloc((size_t)(m + 2), sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) off[slot[kk[e]] + 1]++;
    for (int64_t k = 0; k < m; ++k) off[k + 1] += off[k];
    int64_t *ord = (int64_t *)malloc(sizeof(int64_t) * (size_t)(E + 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    for (int64_t k = 0; k < m; ++k) cur[k] = off[k];
    for (int64_t e = 0; e < E; ++e) ord[cur[slot[kk[e]]]++] = e;
    REAL *Ek = (REAL *)malloc(sizeof(REAL) * (size_t)(D + 1));   /* dense column, reused */
    int64_t *cams = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    char *seen = (char *)calloc((size_t)(n + 1), 1);
    for (int64_t i = 0; i < D; ++i) Ek[i] = 0;

    REAL *S = B, *y = g;   /* accumulate the Schur complement in place */
    REAL *dX = (REAL *)calloc((size_t)(D + 1), sizeof(REAL));
    int failed = 0;
    int solve = !(structure_only || n == 0);
    REAL *dZ = (REAL *)malloc(sizeof(REAL) * (size_t)(m + 1));

    for (int pass = 0; pass < 2; ++pass) {
        /* pass 0: S, y and solve ; pass 1: back-substitution dZ = Q (w - E^T dX) */
        if (pass == 0 && !solve) continue;
        for (int64_t k = 0; k < m; ++k) {
            int64_t nc = 0;
            for (int64_t s = off[k]; s < off[k + 1]; ++s) {
                int64_t e = ord[s], a = ii[e] - fixedp, b = jj[e] - fixedp;
                if (a >= 0) { if (!seen[a]) { seen[a] = 1; cams[nc++] = a; } for (int p = 0; p < 6; ++p) Ek[6*a + p] += Ei[6*e + p]; }
                if (b >= 0) { if (!seen[b]) { seen[b] = 1; cams[nc++] = b; } for (int p = 0; p < 6; ++p) Ek[6*b + p] += Ej[6*e + p]; }
            }
            if (pass == 0) {
                for (int64_t u = 0; u < nc; ++u) for (int64_t v = 0; v < nc; ++v)
                    for (int p = 0; p < 6; ++p) for (int q = 0; q < 6; ++q)
                        S[(6*cams[u] + p)*D + 6*cams[v] + q] -= (Ek[6*cams[u] + p]*Q[k]) * Ek[6*cams[v] + q];
                for (int64_t u = 0; u < nc; ++u) for (int p = 0; p < 6; ++p)
                    y[6*cams[u] + p] -= (Ek[6*cams[u] + p]*Q[k]) * w[k];
            } else {
                REAL acc = 0;
                for (int64_t u = 0; u < nc; ++u) for (int p = 0; p < 6; ++p)
                    acc += Ek[6*cams[u] + p] * dX[6*cams[u] + p];
                dZ[k] = Q[k] * (w[k] - acc);
            }
            for (int64_t u = 0; u < nc; ++u) { seen[cams[u]] = 0; for (int p = 0; p < 6; ++p) Ek[6*cams[u] + p] = 0; }
        }
        if (pass == 0) {
            if (S_out) memcpy(S_out, S, sizeof(REAL) * (size_t)(D * D));
            if (y_out) memcpy(y_out, y, sizeof(REAL) * (size_t)D);
            REAL *A = (REAL *)malloc(sizeof(REAL) * (size_t)(D * D + 1));
            REAL lms[2] = { (REAL)1e-4, (REAL)1e-3 };                      /* ba.py:323-325 */
            for (int attempt = 0; attempt < 2; ++attempt) {
                memcpy(A, S, sizeof(REAL) * (size_t)(D * D));
                for (int64_t i = 0; i < D; ++i) A[i*D + i] = A[i*D + i] + (ep + lms[attempt]*A[i*D + i]);   /* ba.py:67 */
                memcpy(dX, y, sizeof(REAL) * (size_t)D);
                failed = FN(chol_solve)(A, dX, (int)D);
                if (failed) { for (int64_t i = 0; i < D; ++i) dX[i] = 0; }
                int has_nan = 0;
                for (int64_t i = 0; i < D; ++i) if (dX[i] != dX[i]) has_nan = 1;
                if (!has_nan) break;
            }
            free(A);
            if (dX_out) memcpy(dX_out, dX, sizeof(REAL) * (size_t)D);
        }
    }
    if (!solve) for (int64_t k = 0; k < m; ++k) dZ[k] = Q[k] * w[k];      /* ba.py:316-317 */

    /* retraction                                                       ba.py:332-337 */
    for (int64_t p = 0; p < P_tot; ++p) {
        REAL dd = patches[3*p + 2] + (slot[p] >= 0 ? dZ[slot[p]] : (REAL)0);
        dd = dd < (REAL)1e-3 ? (REAL)1e-3 : dd; dd = dd > (REAL)10 ? (REAL)10 : dd;
        patches_out[3*p] = patches[3*p]; patches_out[3*p + 1] = patches[3*p + 1]; patches_out[3*p + 2] = dd;
    }
    if (solve) {
        for (int64_t p = 0; p < N_buf; ++p) {
            REAL xi[6] = { 0, 0, 0, 0, 0, 0 };
            if (p >= fixedp && p < fixedp + n) for (int c = 0; c < 6; ++c) xi[c] = dX[6*(p - fixedp) + c];
            FN(se3) r = FN(se3_mul)(FN(se3_exp)(xi), FN(se3_load)(poses + 7*p));   /* groups.py:153-156 */
            for (int c = 0; c < 3; ++c) poses_out[7*p + c] = r.t[c];
            for (int c = 0; c < 4; ++c) poses_out[7*p + 3 + c] = r.q[c];
        }
    } else {
        memcpy(poses_out, poses, sizeof(REAL) * (size_t)(7 * N_buf));
    }

    free(slot); free(kx); free(B); free(g); free(C); free(w); free(Ei); free(Ej); free(Q);
    free(off); free(ord); free(cur); free(Ek); free(cams); free(seen); free(dX); free(dZ);
    return failed ? 1 : 0;
}

#undef FN
#undef CAT
#undef CAT_
