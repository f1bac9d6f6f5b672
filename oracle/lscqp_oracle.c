/*
 * lscqp_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lscqp_oracle.h for scope, the
 * reference lines each function follows, and how the oracle is pinned.
 *
 * Build: gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC lscqp_oracle.c -o liblscqp_oracle.so -lm
 */
#include "lscqp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS_FLOAT 1e-5 /* SP_EPSILON_FLOAT, include/sp_const.hpp:4 */
#define ORC_EPS 1e-9       /* SP_EPSILON,       include/sp_const.hpp:3 */

/* ------------------------------------------------------------------------------------------------
 * include/polynomial.hpp:9-20
 * ---------------------------------------------------------------------------------------------- */
static int n_choose_k(int n, int k) {
    if (k > n) return 0;
    if (k * 2 > n) k = n - k;
    if (k == 0) return 1;
    int result = n;
    for (int i = 2; i <= k; i++) {
        result *= (n - i + 1);
        result /= i;
    }
    return result;
}

/* include/polynomial.hpp:90-100 */
static int coef_derivative(int n, int phi) {
    if (n < phi) return 0;
    int coef = 1;
    for (int i = 0; i < phi; i++) coef *= n - i;
    return coef;
}

/* include/polynomial.hpp:281-294 — B(i,j) = C(n,i) C(n-i,n-j) (-1)^(j-i) for j >= i */
void orc_bernstein(int n, double* B) {
    int N = n + 1;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            B[i * N + j] = (j >= i) ? n_choose_k(n, i) * n_choose_k(n - i, n - j) * (((j - i) & 1) ? -1.0 : 1.0) : 0.0;
}

/* src/traj_optimizer.cpp:163-178 */
void orc_q_base(int n, int phi, int phi_n, double dt, double* Q) {
    int N = n + 1;
    double* B = (double*)malloc(sizeof(double) * N * N);
    double* Z = (double*)malloc(sizeof(double) * N * N);
    double* T = (double*)malloc(sizeof(double) * N * N);
    orc_bernstein(n, B);
    memset(Q, 0, sizeof(double) * N * N);
    for (int k = phi; k > phi - phi_n; k--) {
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                Z[i * N + j] = 0.0;
                if (i + j - 2 * k + 1 > 0)
                    Z[i * N + j] = (double)coef_derivative(i, k) * coef_derivative(j, k) / (i + j - 2 * k + 1);
            }
        /* T = B * Z */
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int l = 0; l < N; l++) s += B[i * N + l] * Z[l * N + j];
                T[i * N + j] = s;
            }
        /* Q += (T * B^T) * dt^(-2k+1) */
        double sc = pow(dt, -2 * k + 1);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int l = 0; l < N; l++) s += T[i * N + l] * B[j * N + l];
                Q[i * N + j] += s * sc;
            }
    }
    free(B);
    free(Z);
    free(T);
}

/* src/traj_optimizer.cpp:180-214 */
int orc_aeq_base(int M, int n, int phi, double dt, double* Aeq) {
    if (!(n == 5 && phi == 3)) return -1;
    static const double A_0[6][6] = {{1, 0, 0, 0, 0, 0},   {-1, 1, 0, 0, 0, 0},  {1, -2, 1, 0, 0, 0},
                                     {-1, 3, -3, 1, 0, 0}, {1, -4, 6, -4, 1, 0}, {-1, 5, -10, 10, -5, 1}};
    static const double A_T[6][6] = {{0, 0, 0, 0, 0, 1},   {0, 0, 0, 0, -1, 1},  {0, 0, 0, 1, -2, 1},
                                     {0, 0, -1, 3, -3, 1}, {0, 1, -4, 6, -4, 1}, {-1, 5, -10, 10, -5, 1}};
    int P = M * (n + 1);
    int rows = (M - 2) * phi;
    if (rows > 0) memset(Aeq, 0, sizeof(double) * rows * P);
    for (int m = 2; m < M; m++) {
        int nn = 1;
        for (int j = 0; j < phi; j++) {
            double sc = pow(dt, -j) * nn;
            for (int i = 0; i < n + 1; i++) {
                Aeq[(phi * (m - 2) + j) * P + (n + 1) * (m - 1) + i] = sc * A_T[j][i];
                Aeq[(phi * (m - 2) + j) * P + (n + 1) * m + i] = -sc * A_0[j][i];
            }
            nn = nn * (n - j);
        }
    }
    return 0;
}

/* src/traj_optimizer.cpp:530-538.  point3d arithmetic is float32 in octomap (operator- and norm_sq on
 * floats, sqrt in double); mimic it. */
int orc_terminal_segments(const orc_class* c, const orc_agent* a) {
    float dx = (float)a->goal[0] - (float)a->p0[0];
    float dy = (float)a->goal[1] - (float)a->p0[1];
    float dz = (float)a->goal[2] - (float)a->p0[2];
    float nsq = dx * dx + dy * dy + dz * dz;
    double ideal_flight_time = sqrt((double)nsq) / a->nominal_velocity;
    int ts = (int)((c->M * c->dt - ideal_flight_time + ORC_EPS) / c->dt);
    return ts > 1 ? ts : 1;
}

static int lsc_row_used(const orc_class* c, const orc_lsc* l, int m, int i) {
    if (m == 0 && i < c->phi) return 0; /* :404-406 */
    double nn = sqrt(l->nrm[0] * l->nrm[0] + l->nrm[1] * l->nrm[1] + l->nrm[2] * l->nrm[2]);
    if (nn < ORC_EPS_FLOAT) return -1; /* :409-411 */
    return 1;
}

void orc_count(const orc_class* c, const orc_agent* a, const orc_lsc* lsc, orc_sizes* out) {
    int M = c->M, n = c->n, phi = c->phi, dim = c->dim;
    int P = M * (n + 1);
    memset(out, 0, sizeof(*out));
    out->nv = dim * P;
    /* :318-353 six rows per axis (five if M == 1), :356-368, :502-511 */
    out->neq = dim * (M > 1 ? 6 : 5) + dim * (M - 2 > 0 ? (M - 2) * phi : 0) + (c->planner_lsc == 1 ? dim * (phi - 1) : 0);
    if (c->use_sfc) out->n_sfc = 2 * dim * (P - phi);
    for (int oi = 0; oi < a->n_obs; oi++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n + 1; i++) {
                int u = lsc_row_used(c, &lsc[(oi * M + m) * (n + 1) + i], m, i);
                if (u == 1) out->n_lsc++;
                if (u == -1) out->n_lsc_skipped++;
            }
    out->n_vel = 2 * dim * (M * n - 2);
    out->n_acc = 2 * dim * (M * (n - 1) - 1);
    if (c->comm_range > 0) out->n_comm = 2 * dim * (M * (M + 1) / 2 + M);
    out->nineq = out->n_sfc + out->n_lsc + out->n_vel + out->n_acc + out->n_comm;
}

/* src/traj_optimizer.cpp:216-514, row for row and in the same order.  Rows the reference writes as
 * "expr >= 0" are negated into "G x <= h". */
void orc_assemble(const orc_class* c, const orc_agent* a, const orc_lsc* lsc, const orc_box* sfc, double* P_,
                  double* q, double* r, double* Aeq, double* beq, double* G, double* h, double* lb, double* ub) {
    int M = c->M, n = c->n, phi = c->phi, dim = c->dim;
    int offset_seg = n + 1, offset_dim = M * (n + 1);
    int nv = dim * offset_dim;
    orc_sizes sz;
    orc_count(c, a, lsc, &sz);
    double dt = c->dt;

    /* variables and bounds, :238-270 */
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n + 1; i++) {
                int row = k * offset_dim + m * offset_seg + i;
                if (m == 0 && i < 3) {
                    lb[row] = -INFINITY;
                    ub[row] = INFINITY;
                } else {
                    lb[row] = c->world_min[k];
                    ub[row] = c->world_max[k];
                    if (k == 2 && m == 0 && c->planner_lsc == 2) { /* RECIPROCALRSFC, "to avoid numerical error" (:255-258) */
                        lb[row] = -100;
                        ub[row] = 100;
                    }
                }
            }

    /* cost, :285-316 */
    double Qb[36];
    orc_q_base(n, phi, c->phi_n, dt, Qb);
    memset(P_, 0, sizeof(double) * nv * nv);
    memset(q, 0, sizeof(double) * nv);
    *r = 0;
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n + 1; i++) {
                int row = k * offset_dim + m * offset_seg + i;
                for (int j = 0; j < n + 1; j++) {
                    int col = k * offset_dim + m * offset_seg + j;
                    if (Qb[i * (n + 1) + j] != 0 && c->w_c != 0) P_[row * nv + col] += c->w_c * Qb[i * (n + 1) + j];
                }
            }
    int ts = orc_terminal_segments(c, a);
    for (int m = M - ts; m < M; m++)
        for (int k = 0; k < dim; k++) {
            int v = k * offset_dim + m * offset_seg + n;
            P_[v * nv + v] += c->w_t;
            q[v] += -2.0 * c->w_t * a->goal[k];
            *r += c->w_t * a->goal[k] * a->goal[k];
        }

    /* equalities */
    memset(Aeq, 0, sizeof(double) * sz.neq * nv);
    memset(beq, 0, sizeof(double) * sz.neq);
    int e = 0;
    for (int k = 0; k < dim; k++) { /* :319-353 */
        int b0 = k * offset_dim;
        Aeq[e * nv + b0 + 0] = 1;
        beq[e] = a->p0[k];
        e++;
        if (M > 1) {
            Aeq[e * nv + b0 + n] = 1;
            Aeq[e * nv + b0 + offset_seg] = -1;
            e++;
        }
        double s1 = pow(dt, -1) * n;
        Aeq[e * nv + b0 + 1] = s1;
        Aeq[e * nv + b0 + 0] = -s1;
        beq[e] = a->v0[k];
        e++;
        double s2 = pow(dt, -2) * n * (n - 1);
        Aeq[e * nv + b0 + 2] = s2;
        Aeq[e * nv + b0 + 1] = -2 * s2;
        Aeq[e * nv + b0 + 0] = s2;
        beq[e] = a->a0[k];
        e++;
        /* back velocity */
        Aeq[e * nv + b0 + offset_seg + 1] += 1;
        Aeq[e * nv + b0 + offset_seg + 0] += -1;
        Aeq[e * nv + b0 + n] += -1;
        Aeq[e * nv + b0 + n - 1] += 1;
        e++;
        /* back acceleration */
        Aeq[e * nv + b0 + offset_seg + 2] += 1;
        Aeq[e * nv + b0 + offset_seg + 1] += -2;
        Aeq[e * nv + b0 + offset_seg + 0] += 1;
        Aeq[e * nv + b0 + n] += -1;
        Aeq[e * nv + b0 + n - 1] += 2;
        Aeq[e * nv + b0 + n - 2] += -1;
        e++;
    }
    if (M > 2) { /* :357-368 */
        int rows = (M - 2) * phi;
        double* Ab = (double*)malloc(sizeof(double) * rows * offset_dim);
        orc_aeq_base(M, n, phi, dt, Ab);
        for (int k = 0; k < dim; k++)
            for (int i = 0; i < rows; i++) {
                for (int j = 0; j < offset_dim; j++)
                    if (Ab[i * offset_dim + j] != 0) Aeq[e * nv + k * offset_dim + j] = Ab[i * offset_dim + j];
                e++;
            }
        free(Ab);
    }
    if (c->planner_lsc == 1) { /* :504-511 */
        for (int k = 0; k < dim; k++) {
            int m = M - 1;
            for (int i = 1; i < phi; i++) {
                Aeq[e * nv + k * offset_dim + m * offset_seg + n] = 1;
                Aeq[e * nv + k * offset_dim + m * offset_seg + n - i] = -1;
                e++;
            }
        }
    }

    /* inequalities */
    memset(G, 0, sizeof(double) * sz.nineq * nv);
    memset(h, 0, sizeof(double) * sz.nineq);
    int g = 0;
    if (c->use_sfc) { /* :372-397 with Box::convertToLSCs (src/collision_constraints.cpp:37-59) */
        for (int m = 0; m < M; m++) {
            for (int f = 0; f < 2 * dim; f++) {
                int ax = f / 2;
                double nrm = (f % 2 == 0) ? 1.0 : -1.0;
                double d = (f % 2 == 0) ? sfc[m].bmin[ax] : -sfc[m].bmax[ax];
                for (int j = 0; j < n + 1; j++) {
                    if (m == 0 && j < phi) continue;
                    /* nrm * x - d >= 0  ->  -nrm * x <= -d */
                    G[g * nv + ax * offset_dim + m * offset_seg + j] = -nrm;
                    h[g] = -d;
                    g++;
                }
            }
        }
    }
    for (int oi = 0; oi < a->n_obs; oi++) /* :401-437 */
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n + 1; i++) {
                const orc_lsc* l = &lsc[(oi * M + m) * (n + 1) + i];
                if (lsc_row_used(c, l, m, i) != 1) continue;
                double rhs = l->d;
                for (int k = 0; k < dim; k++) {
                    G[g * nv + k * offset_dim + m * offset_seg + i] = -l->nrm[k];
                    rhs += l->nrm[k] * l->p[k];
                }
                h[g] = -rhs;
                g++;
            }
    for (int k = 0; k < dim; k++) /* :440-474 */
        for (int m = 0; m < M; m++) {
            double s1 = pow(dt, -1) * n;
            for (int i = 0; i < n; i++) {
                if (m == 0 && (i == 0 || i == 1)) continue;
                int v = k * offset_dim + m * offset_seg + i;
                G[g * nv + v + 1] = s1;
                G[g * nv + v] = -s1;
                h[g] = a->vmax[k];
                g++;
                G[g * nv + v + 1] = -s1;
                G[g * nv + v] = s1;
                h[g] = a->vmax[k];
                g++;
            }
            double s2 = pow(dt, -2) * n * (n - 1);
            for (int i = 0; i < n - 1; i++) {
                if (m == 0 && i == 0) continue;
                int v = k * offset_dim + m * offset_seg + i;
                G[g * nv + v + 2] = s2;
                G[g * nv + v + 1] = -2 * s2;
                G[g * nv + v] = s2;
                h[g] = a->amax[k];
                g++;
                G[g * nv + v + 2] = -s2;
                G[g * nv + v + 1] = 2 * s2;
                G[g * nv + v] = -s2;
                h[g] = a->amax[k];
                g++;
            }
        }
    if (c->comm_range > 0) { /* :478-500 */
        for (int k = 0; k < dim; k++)
            for (int mi = 0; mi < M; mi++)
                for (int m = mi; m < M; m++) {
                    int va = k * offset_dim + m * offset_seg + n;
                    int vb = k * offset_dim + mi * offset_seg + 0;
                    G[g * nv + va] += 1;
                    G[g * nv + vb] += -1;
                    h[g] = 0.5 * c->comm_range - a->radius;
                    g++;
                    G[g * nv + va] += -1;
                    G[g * nv + vb] += 1;
                    h[g] = 0.5 * c->comm_range - a->radius;
                    g++;
                }
        for (int k = 0; k < dim; k++)
            for (int m = 0; m < M; m++) {
                int va = k * offset_dim + m * offset_seg + n;
                G[g * nv + va] = 1;
                h[g] = 0.5 * c->comm_range - ORC_EPS_FLOAT + a->next_waypoint[k];
                g++;
                G[g * nv + va] = -1;
                h[g] = 0.5 * c->comm_range - ORC_EPS_FLOAT - a->next_waypoint[k];
                g++;
            }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Dense linear algebra helpers (row-major)
 * ---------------------------------------------------------------------------------------------- */
static int chol_factor(int n, double* A) { /* lower, in place; returns 0 ok */
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(int n, const double* L, double* b) {
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}

/* Householder QR of A^T where A is (neq x nv): returns Q (nv x nv, row-major, explicit) and R (neq x neq upper). */
static void qr_of_transpose(int neq, int nv, const double* A, double* Q, double* R) {
    /* W = A^T, nv x neq */
    double* W = (double*)malloc(sizeof(double) * nv * neq);
    double* v = (double*)malloc(sizeof(double) * nv);
    for (int i = 0; i < nv; i++)
        for (int j = 0; j < neq; j++) W[i * neq + j] = A[j * nv + i];
    for (int i = 0; i < nv; i++)
        for (int j = 0; j < nv; j++) Q[i * nv + j] = (i == j) ? 1.0 : 0.0;
    for (int j = 0; j < neq; j++) {
        double nrm = 0;
        for (int i = j; i < nv; i++) nrm += W[i * neq + j] * W[i * neq + j];
        nrm = sqrt(nrm);
        if (nrm == 0) continue;
        double alpha = (W[j * neq + j] > 0) ? -nrm : nrm;
        for (int i = 0; i < nv; i++) v[i] = 0;
        for (int i = j; i < nv; i++) v[i] = W[i * neq + j];
        v[j] -= alpha;
        double vn = 0;
        for (int i = j; i < nv; i++) vn += v[i] * v[i];
        if (vn == 0) continue;
        /* W = (I - 2 v v^T / vn) W */
        for (int cidx = j; cidx < neq; cidx++) {
            double s = 0;
            for (int i = j; i < nv; i++) s += v[i] * W[i * neq + cidx];
            s = 2 * s / vn;
            for (int i = j; i < nv; i++) W[i * neq + cidx] -= s * v[i];
        }
        /* Q = Q (I - 2 v v^T / vn) */
        for (int i = 0; i < nv; i++) {
            double s = 0;
            for (int l = j; l < nv; l++) s += Q[i * nv + l] * v[l];
            s = 2 * s / vn;
            for (int l = j; l < nv; l++) Q[i * nv + l] -= s * v[l];
        }
    }
    for (int i = 0; i < neq; i++)
        for (int j = 0; j < neq; j++) R[i * neq + j] = (j >= i) ? W[i * neq + j] : 0.0;
    free(W);
    free(v);
}

/* ------------------------------------------------------------------------------------------------
 * Generic dense Mehrotra predictor-corrector:  min 1/2 z'Hz + g'z  s.t.  Gz <= h
 * ---------------------------------------------------------------------------------------------- */
/* weighted != 0 (second attempt of orc_solve only, round 4): in iterations whose affine step is blocked below 0.3 while the primal residual
 * has closed, Mehrotra's second-order term ds_a dl_a -- the error of a FULL affine step -- is weighted with that step length; the plain
 * iteration can run into a limit cycle there (the same finding as in the HIP kernel: NOTES.md section 11, "A limit cycle ...").  The first
 * attempt is the iteration every golden vector was pinned with, unchanged. */
static int pdip_dense(int nz, int m, const double* H, const double* g, const double* G, const double* h,
                      const double* z0, double tol, int max_iter, double* z, double* lam, int* iters_out, int weighted) {
    double* s = (double*)malloc(sizeof(double) * m);
    double* w = (double*)malloc(sizeof(double) * m);
    double* rp = (double*)malloc(sizeof(double) * m);
    double* t = (double*)malloc(sizeof(double) * m);
    double* ds = (double*)malloc(sizeof(double) * m);
    double* dl = (double*)malloc(sizeof(double) * m);
    double* dsa = (double*)malloc(sizeof(double) * m);
    double* dla = (double*)malloc(sizeof(double) * m);
    double* rd = (double*)malloc(sizeof(double) * nz);
    double* rhs = (double*)malloc(sizeof(double) * nz);
    double* dz = (double*)malloc(sizeof(double) * nz);
    double* K = (double*)malloc(sizeof(double) * nz * nz);
    int status = 2, it = 0;
    double gscale = 1.0, hscale = 1.0;
    for (int i = 0; i < nz; i++) gscale = fmax(gscale, fabs(g[i]));
    for (int i = 0; i < m; i++)
        if (isfinite(h[i])) hscale = fmax(hscale, fabs(h[i]));

    for (int i = 0; i < nz; i++) z[i] = z0 ? z0[i] : 0.0;
    for (int i = 0; i < m; i++) {
        double gz = 0;
        for (int j = 0; j < nz; j++) gz += G[i * nz + j] * z[j];
        double si = h[i] - gz;
        s[i] = si > 1.0 ? si : 1.0;
        lam[i] = 1.0;
    }
    int stall = 0, near_opt = 0, floor_seen = 0;
    for (it = 0; it < max_iter; it++) {
        /* residuals */
        double mu = 0, rpn = 0, rdn = 0, pinf = 0;
        for (int i = 0; i < m; i++) {
            double gz = 0;
            for (int j = 0; j < nz; j++) gz += G[i * nz + j] * z[j];
            rp[i] = gz + s[i] - h[i];
            rpn = fmax(rpn, fabs(rp[i]));
            mu += s[i] * lam[i];
            pinf += lam[i] * fabs(rp[i]); /* objective-unit weight of the primal residual */
        }
        mu = m > 0 ? mu / m : 0.0;
        double objv = 0;
        for (int i = 0; i < nz; i++) {
            double hz = 0;
            for (int j = 0; j < nz; j++) hz += H[i * nz + j] * z[j];
            objv += z[i] * (0.5 * hz + g[i]);
            rd[i] = hz + g[i];
        }
        double gls = gscale;
        for (int i = 0; i < nz; i++) gls = fmax(gls, fabs(rd[i])); /* |Hz + g| before the multiplier term */
        for (int i = 0; i < m; i++) {
            double li = lam[i];
            for (int j = 0; j < nz; j++) rd[j] += G[i * nz + j] * li;
        }
        for (int i = 0; i < nz; i++) rdn = fmax(rdn, fabs(rd[i]));
        if (rpn <= tol * hscale && rdn <= 10 * tol * gls && mu * m + pinf <= tol * (1.0 + fabs(objv))) {
            status = 0;
            break;
        }
        /* "converged to working precision": every criterion within 100x of its target.  Remembered so that a later
         * factorisation failure or a stalled step (W = lam/s spans > 1e20 by then) still reports the optimum. */
        near_opt = (rpn <= 100 * tol * hscale && rdn <= 1000 * tol * gls && mu * m + pinf <= 100 * tol * (1.0 + fabs(objv)));
        /* The stationarity residual has a rounding floor of ~eps * cond(K) * |grad|; at M = 10 in 3-D (nz = 84, cond ~ 1e7)
         * it can sit above 1000 tol.  A point that meets the primal and gap targets with the stationarity below 1e-6 is
         * remembered too and accepted if the iteration later breaks down (sticky, unlike near_opt). */
        if (rpn <= 100 * tol * hscale && rdn <= 1e-6 * gls && mu * m + pinf <= 100 * tol * (1.0 + fabs(objv))) floor_seen = 1;
        /* K = H + G' W G */
        memcpy(K, H, sizeof(double) * nz * nz);
        for (int i = 0; i < m; i++) {
            w[i] = lam[i] / s[i];
            const double* gi = &G[i * nz];
            for (int a = 0; a < nz; a++) {
                double ga = gi[a];
                if (ga == 0) continue;
                double wa = w[i] * ga;
                for (int b = 0; b <= a; b++) K[a * nz + b] += wa * gi[b];
            }
        }
        for (int a = 0; a < nz; a++)
            for (int b = a + 1; b < nz; b++) K[a * nz + b] = K[b * nz + a];
        if (chol_factor(nz, K) != 0) {
            status = (near_opt || floor_seen) ? 0 : 3;
            break;
        }
        /* affine: rc = s*lam */
        for (int i = 0; i < m; i++) t[i] = lam[i] - w[i] * rp[i]; /* rc/s - W rp with rc = s lam */
        for (int j = 0; j < nz; j++) rhs[j] = -rd[j];
        for (int i = 0; i < m; i++) {
            double ti = t[i];
            for (int j = 0; j < nz; j++) rhs[j] += G[i * nz + j] * ti;
        }
        memcpy(dz, rhs, sizeof(double) * nz);
        chol_solve(nz, K, dz);
        double alpha_aff = 1.0;
        for (int i = 0; i < m; i++) {
            double gdz = 0;
            for (int j = 0; j < nz; j++) gdz += G[i * nz + j] * dz[j];
            dsa[i] = -rp[i] - gdz;
            dla[i] = -lam[i] + w[i] * (rp[i] + gdz);
            if (dsa[i] < 0) alpha_aff = fmin(alpha_aff, -s[i] / dsa[i]);
            if (dla[i] < 0) alpha_aff = fmin(alpha_aff, -lam[i] / dla[i]);
        }
        double mu_aff = 0;
        for (int i = 0; i < m; i++) mu_aff += (s[i] + alpha_aff * dsa[i]) * (lam[i] + alpha_aff * dla[i]);
        mu_aff = m > 0 ? mu_aff / m : 0.0;
        double sigma = mu > 0 ? pow(mu_aff / mu, 3.0) : 0.0;
        /* corrector: rc = s*lam - sigma*mu + dsa*dla */
        if (weighted && alpha_aff < 0.3 && rpn <= 1e-6 * hscale)
            for (int i = 0; i < m; i++) dsa[i] *= alpha_aff; /* (dsa is only used in the product from here on) */
        for (int i = 0; i < m; i++) t[i] = (s[i] * lam[i] - sigma * mu + dsa[i] * dla[i]) / s[i] - w[i] * rp[i];
        for (int j = 0; j < nz; j++) rhs[j] = -rd[j];
        for (int i = 0; i < m; i++) {
            double ti = t[i];
            for (int j = 0; j < nz; j++) rhs[j] += G[i * nz + j] * ti;
        }
        memcpy(dz, rhs, sizeof(double) * nz);
        chol_solve(nz, K, dz);
        double alpha = 1.0, amax = 1e300;
        for (int i = 0; i < m; i++) {
            double gdz = 0;
            for (int j = 0; j < nz; j++) gdz += G[i * nz + j] * dz[j];
            ds[i] = -rp[i] - gdz;
            dl[i] = -(s[i] * lam[i] - sigma * mu + dsa[i] * dla[i]) / s[i] + w[i] * (rp[i] + gdz);
            if (ds[i] < 0) amax = fmin(amax, -s[i] / ds[i]);
            if (dl[i] < 0) amax = fmin(amax, -lam[i] / dl[i]);
        }
        alpha = fmin(1.0, 0.995 * amax);
        /* centrality safeguard: no complementarity product below 1e-4 mu along the step (without it Mehrotra's heuristic
         * can cycle with mu ~ 1e-9 at M = 10 in 3-D, see tools/proto_pdip.py) */
        for (int bt = 0; bt < 10; bt++) {
            double mua = 0, pmin = 1e300;
            for (int i = 0; i < m; i++) {
                const double pr = (s[i] + alpha * ds[i]) * (lam[i] + alpha * dl[i]);
                mua += pr;
                pmin = fmin(pmin, pr);
            }
            if (m == 0 || pmin >= 1e-4 * mua / m) break;
            alpha *= 0.7;
        }
        for (int j = 0; j < nz; j++) z[j] += alpha * dz[j];
        for (int i = 0; i < m; i++) {
            s[i] += alpha * ds[i];
            lam[i] += alpha * dl[i];
        }
        if (alpha < 1e-10) {
            if (++stall >= 3) {
                status = (near_opt || floor_seen) ? 0 : (rpn > 1e-6 * hscale) ? 1 : 3;
                break;
            }
        } else
            stall = 0;
    }
    if (status == 2 && (near_opt || floor_seen)) status = 0;
    if (status == 2) {
        /* iteration limit: classify as infeasible when the primal residual never closed */
        double rpn = 0;
        for (int i = 0; i < m; i++) {
            double gz = 0;
            for (int j = 0; j < nz; j++) gz += G[i * nz + j] * z[j];
            rpn = fmax(rpn, gz - h[i]);
        }
        if (rpn > 1e-6 * hscale) status = 1;
    }
    if (iters_out) *iters_out = it;
    free(s); free(w); free(rp); free(t); free(ds); free(dl); free(dsa); free(dla);
    free(rd); free(rhs); free(dz); free(K);
    return status;
}

int orc_solve(const orc_class* c, const orc_agent* a, const orc_lsc* lsc, const orc_box* sfc, double tol,
              int max_iter, double* x, double* obj, double* y, double* lam, double* mu_lb, double* mu_ub,
              int* iters) {
    orc_sizes sz;
    orc_count(c, a, lsc, &sz);
    int nv = sz.nv, neq = sz.neq, mi = sz.nineq;
    if (tol <= 0) tol = 1e-11;
    if (max_iter <= 0) max_iter = 200;
    double* P = (double*)malloc(sizeof(double) * nv * nv);
    double* q = (double*)malloc(sizeof(double) * nv);
    double* Aeq = (double*)malloc(sizeof(double) * (neq > 0 ? neq : 1) * nv);
    double* beq = (double*)malloc(sizeof(double) * (neq > 0 ? neq : 1));
    double* G = (double*)malloc(sizeof(double) * (mi > 0 ? mi : 1) * nv);
    double* h = (double*)malloc(sizeof(double) * (mi > 0 ? mi : 1));
    double* lb = (double*)malloc(sizeof(double) * nv);
    double* ub = (double*)malloc(sizeof(double) * nv);
    double r;
    orc_assemble(c, a, lsc, sfc, P, q, &r, Aeq, beq, G, h, lb, ub);

    /* Solve in coordinates translated by p0 (x = x' + t): the same problem, but every quantity is O(1 m),
     * which keeps the rounding floor of the stationarity residual low.  Undone before returning. */
    double* tsh = (double*)malloc(sizeof(double) * nv);
    double* q_orig = (double*)malloc(sizeof(double) * nv); /* untranslated linear term, for the reported objective */
    memcpy(q_orig, q, sizeof(double) * nv);
    {
        int Pn = c->M * (c->n + 1);
        for (int i = 0; i < nv; i++) tsh[i] = a->p0[i / Pn];
        for (int e = 0; e < neq; e++) {
            double s = 0;
            for (int l = 0; l < nv; l++) s += Aeq[e * nv + l] * tsh[l];
            beq[e] -= s;
        }
        for (int i = 0; i < mi; i++) {
            double s = 0;
            for (int l = 0; l < nv; l++) s += G[i * nv + l] * tsh[l];
            h[i] -= s;
        }
        for (int i = 0; i < nv; i++) {
            double s = 0;
            for (int l = 0; l < nv; l++) s += 2 * P[i * nv + l] * tsh[l];
            q[i] += s; /* the constant term is not needed: the objective is re-evaluated on x below */
            lb[i] -= tsh[i];
            ub[i] -= tsh[i];
        }
    }

    /* bounds -> rows */
    int nb = 0;
    for (int i = 0; i < nv; i++) nb += (isfinite(lb[i]) ? 1 : 0) + (isfinite(ub[i]) ? 1 : 0);
    int m = mi + nb;
    int nz = nv - neq;

    double* Q = (double*)malloc(sizeof(double) * nv * nv);
    double* R = (double*)malloc(sizeof(double) * neq * neq);
    qr_of_transpose(neq, nv, Aeq, Q, R);
    /* x_p = Y R^{-T} beq  with  Aeq = R^T Y^T */
    double* u = (double*)malloc(sizeof(double) * neq);
    for (int i = 0; i < neq; i++) {
        double s = beq[i];
        for (int k = 0; k < i; k++) s -= R[k * neq + i] * u[k];
        u[i] = s / R[i * neq + i];
    }
    double* xp = (double*)calloc(nv, sizeof(double));
    for (int i = 0; i < nv; i++)
        for (int k = 0; k < neq; k++) xp[i] += Q[i * nv + k] * u[k];
#define NMAT(i, j) Q[(i) * nv + neq + (j)]
    /* Hz = N'(2P)N, gz = N'(2P xp + q) */
    double* PN = (double*)malloc(sizeof(double) * nv * nz);
    for (int i = 0; i < nv; i++)
        for (int j = 0; j < nz; j++) {
            double s = 0;
            for (int l = 0; l < nv; l++)
                if (P[i * nv + l] != 0) s += P[i * nv + l] * NMAT(l, j);
            PN[i * nz + j] = 2 * s;
        }
    double* Hz = (double*)malloc(sizeof(double) * nz * nz);
    for (int i = 0; i < nz; i++)
        for (int j = 0; j < nz; j++) {
            double s = 0;
            for (int l = 0; l < nv; l++) s += NMAT(l, i) * PN[l * nz + j];
            Hz[i * nz + j] = s;
        }
    for (int i = 0; i < nz; i++)
        for (int j = i + 1; j < nz; j++) {
            double av = 0.5 * (Hz[i * nz + j] + Hz[j * nz + i]);
            Hz[i * nz + j] = Hz[j * nz + i] = av;
        }
    double* gx = (double*)malloc(sizeof(double) * nv);
    for (int i = 0; i < nv; i++) {
        double s = q[i];
        for (int l = 0; l < nv; l++) s += 2 * P[i * nv + l] * xp[l];
        gx[i] = s;
    }
    double* gz = (double*)calloc(nz, sizeof(double));
    for (int j = 0; j < nz; j++)
        for (int l = 0; l < nv; l++) gz[j] += NMAT(l, j) * gx[l];
    /* Gz, hz including bound rows */
    double* Gz = (double*)calloc((size_t)(m > 0 ? m : 1) * nz, sizeof(double));
    double* hz = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    int* bvar = (int*)malloc(sizeof(int) * (nb > 0 ? nb : 1));
    int* bsgn = (int*)malloc(sizeof(int) * (nb > 0 ? nb : 1));
    for (int i = 0; i < mi; i++) {
        double gxp = 0;
        for (int l = 0; l < nv; l++) {
            double gl = G[i * nv + l];
            if (gl == 0) continue;
            gxp += gl * xp[l];
            for (int j = 0; j < nz; j++) Gz[(size_t)i * nz + j] += gl * NMAT(l, j);
        }
        hz[i] = h[i] - gxp;
    }
    int bi = 0;
    for (int l = 0; l < nv; l++) {
        if (isfinite(lb[l])) { /* -x <= -lb */
            for (int j = 0; j < nz; j++) Gz[(size_t)(mi + bi) * nz + j] = -NMAT(l, j);
            hz[mi + bi] = -lb[l] + xp[l];
            bvar[bi] = l;
            bsgn[bi] = -1;
            bi++;
        }
        if (isfinite(ub[l])) {
            for (int j = 0; j < nz; j++) Gz[(size_t)(mi + bi) * nz + j] = NMAT(l, j);
            hz[mi + bi] = ub[l] - xp[l];
            bvar[bi] = l;
            bsgn[bi] = 1;
            bi++;
        }
    }
    /* start: projection of "hover at p0" onto the null space */
    double* z0 = (double*)calloc(nz, sizeof(double));
    {
        for (int j = 0; j < nz; j++)
            for (int l = 0; l < nv; l++) z0[j] += NMAT(l, j) * (0.0 - xp[l]);
    }
    double* z = (double*)malloc(sizeof(double) * nz);
    double* lall = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    int it = 0;
    int status = pdip_dense(nz, m, Hz, gz, Gz, hz, z0, tol, max_iter, z, lall, &it, 0);
    if (status == 2 || status == 3) { /* iteration limit / breakdown: once more with the weighted corrector (see pdip_dense) */
        int it2 = 0;
        status = pdip_dense(nz, m, Hz, gz, Gz, hz, z0, tol, max_iter, z, lall, &it2, 1);
        it += it2;
    }
    for (int i = 0; i < nv; i++) {
        double s = xp[i];
        for (int j = 0; j < nz; j++) s += NMAT(i, j) * z[j];
        x[i] = s + tsh[i];
    }
    /* objective in the reference's own form x'Px + q'x + r (q restored to the untranslated one) */
    /* Evaluated in 80-bit long double: the literal double evaluation cancels ~1e6-sized terms down to O(0.1)
     * and would add ~1e-9 of pure rounding noise to the value the parity tests compare against. */
    long double ov = r;
    for (int i = 0; i < nv; i++) {
        long double s = 0;
        for (int l = 0; l < nv; l++)
            if (P[i * nv + l] != 0) s += (long double)P[i * nv + l] * x[l];
        ov += (long double)x[i] * (s + (long double)q_orig[i]);
    }
    if (obj) *obj = (double)ov;
    if (lam) memcpy(lam, lall, sizeof(double) * mi);
    if (mu_lb) memset(mu_lb, 0, sizeof(double) * nv);
    if (mu_ub) memset(mu_ub, 0, sizeof(double) * nv);
    for (int b = 0; b < nb; b++) {
        if (bsgn[b] < 0 && mu_lb) mu_lb[bvar[b]] = lall[mi + b];
        if (bsgn[b] > 0 && mu_ub) mu_ub[bvar[b]] = lall[mi + b];
    }
    if (y) {
        /* stationarity: 2Px + q + G'lam - mu_lb + mu_ub + Aeq' y = 0, Aeq' = Y R  ->  R y = -Y'(...) */
        double* gr = (double*)malloc(sizeof(double) * nv);
        for (int i = 0; i < nv; i++) {
            double s = q[i]; /* translated q: 2P(x - t) + q' == 2Px + q */
            for (int l = 0; l < nv; l++) s += 2 * P[i * nv + l] * (x[l] - tsh[l]);
            gr[i] = s;
        }
        for (int i = 0; i < mi; i++)
            for (int l = 0; l < nv; l++)
                if (G[i * nv + l] != 0) gr[l] += G[i * nv + l] * lall[i];
        for (int b = 0; b < nb; b++) gr[bvar[b]] += bsgn[b] * lall[mi + b];
        double* yt = (double*)calloc(neq, sizeof(double));
        for (int k = 0; k < neq; k++)
            for (int l = 0; l < nv; l++) yt[k] -= Q[l * nv + k] * gr[l];
        for (int i = neq - 1; i >= 0; i--) {
            double s = yt[i];
            for (int k = i + 1; k < neq; k++) s -= R[i * neq + k] * y[k];
            y[i] = s / R[i * neq + i];
        }
        free(gr);
        free(yt);
    }
    if (iters) *iters = it;
#undef NMAT
    free(P); free(q); free(Aeq); free(beq); free(G); free(h); free(lb); free(ub);
    free(Q); free(R); free(u); free(xp); free(PN); free(Hz); free(gx); free(gz);
    free(Gz); free(hz); free(bvar); free(bsgn); free(z0); free(z); free(lall); free(tsh); free(q_orig);
    return status;
}

void orc_kkt(const orc_class* c, const orc_agent* a, const orc_lsc* lsc, const orc_box* sfc, const double* x,
             const double* y, const double* lam, const double* mu_lb, const double* mu_ub, double* res) {
    orc_sizes sz;
    orc_count(c, a, lsc, &sz);
    int nv = sz.nv, neq = sz.neq, mi = sz.nineq;
    double* P = (double*)malloc(sizeof(double) * nv * nv);
    double* q = (double*)malloc(sizeof(double) * nv);
    double* Aeq = (double*)malloc(sizeof(double) * (neq > 0 ? neq : 1) * nv);
    double* beq = (double*)malloc(sizeof(double) * (neq > 0 ? neq : 1));
    double* G = (double*)malloc(sizeof(double) * (mi > 0 ? mi : 1) * nv);
    double* h = (double*)malloc(sizeof(double) * (mi > 0 ? mi : 1));
    double* lb = (double*)malloc(sizeof(double) * nv);
    double* ub = (double*)malloc(sizeof(double) * nv);
    double r;
    orc_assemble(c, a, lsc, sfc, P, q, &r, Aeq, beq, G, h, lb, ub);
    double* gr = (double*)malloc(sizeof(double) * nv);
    double gscale = 1.0;
    long double ov = r; /* long double: see orc_solve */
    for (int i = 0; i < nv; i++) {
        long double s = 0;
        for (int l = 0; l < nv; l++) s += (long double)P[i * nv + l] * x[l];
        ov += (long double)x[i] * (s + q[i]);
        gr[i] = (double)(2 * s + q[i]);
        gscale = fmax(gscale, fabs(gr[i]));
    }
    double eqv = 0, iqv = 0, negm = 0, comp = 0;
    for (int e = 0; e < neq; e++) {
        double s = -beq[e];
        for (int l = 0; l < nv; l++) {
            double ae = Aeq[e * nv + l];
            if (ae == 0) continue;
            s += ae * x[l];
            if (y) gr[l] += ae * y[e];
        }
        eqv = fmax(eqv, fabs(s));
    }
    for (int i = 0; i < mi; i++) {
        double s = -h[i];
        for (int l = 0; l < nv; l++) {
            double gl = G[i * nv + l];
            if (gl == 0) continue;
            s += gl * x[l];
            if (lam) gr[l] += gl * lam[i];
        }
        iqv = fmax(iqv, s);
        if (lam) {
            negm = fmax(negm, -lam[i]);
            comp = fmax(comp, fabs(lam[i] * s));
        }
    }
    for (int l = 0; l < nv; l++) {
        if (isfinite(lb[l])) {
            iqv = fmax(iqv, lb[l] - x[l]);
            if (mu_lb) {
                gr[l] -= mu_lb[l];
                negm = fmax(negm, -mu_lb[l]);
                comp = fmax(comp, fabs(mu_lb[l] * (x[l] - lb[l])));
            }
        }
        if (isfinite(ub[l])) {
            iqv = fmax(iqv, x[l] - ub[l]);
            if (mu_ub) {
                gr[l] += mu_ub[l];
                negm = fmax(negm, -mu_ub[l]);
                comp = fmax(comp, fabs(mu_ub[l] * (ub[l] - x[l])));
            }
        }
    }
    double st = 0;
    for (int l = 0; l < nv; l++) st = fmax(st, fabs(gr[l]));
    res[0] = st / gscale;
    res[1] = eqv;
    res[2] = iqv;
    res[3] = negm;
    res[4] = comp;
    res[5] = (double)ov;
    free(P); free(q); free(Aeq); free(beq); free(G); free(h); free(lb); free(ub); free(gr);
}

void orc_primal_check(const orc_class* c, const orc_agent* a, const orc_lsc* lsc, const orc_box* sfc,
                      const double* x, double* obj, double* eq_viol, double* ineq_viol) {
    double res[6];
    orc_kkt(c, a, lsc, sfc, x, NULL, NULL, NULL, NULL, res);
    if (obj) *obj = res[5];
    if (eq_viol) *eq_viol = res[1];
    if (ineq_viol) *ineq_viol = res[2];
}

/* src/trajectory.cpp:111-199 on fp64 control points (the reference evaluates on float32 point3d). */
static double bern_eval(int n, const double* cp, double t) {
    double s = 0;
    for (int i = 0; i < n + 1; i++) s += cp[i] * n_choose_k(n, i) * pow(t, i) * pow(1 - t, n - i);
    return s;
}
void orc_state_at(const orc_class* c, const double* x, double time, double* pos, double* vel, double* acc) {
    int M = c->M, n = c->n, dim = c->dim;
    int P = M * (n + 1);
    int m = -1;
    double tn = 0, end = 0;
    for (int idx = 0; idx < M; idx++) { /* :121-129 */
        end += c->dt;
        if (time < end) {
            m = idx;
            tn = 1 - (end - time) / c->dt;
            break;
        }
    }
    if (m == -1) { /* :131-136 */
        m = M - 1;
        tn = 1.0;
    }
    for (int k = 0; k < dim; k++) {
        const double* cp = &x[k * P + m * (n + 1)];
        double d1[8], d2[8];
        for (int i = 0; i < n; i++) d1[i] = (cp[i + 1] - cp[i]) * (n / c->dt); /* :183-199 */
        for (int i = 0; i < n - 1; i++) d2[i] = (d1[i + 1] - d1[i]) * ((n - 1) / c->dt);
        pos[k] = bern_eval(n, cp, tn);
        vel[k] = bern_eval(n - 1, d1, tn);
        acc[k] = bern_eval(n - 2, d2, tn);
    }
}

int orc_solve_batch(const orc_class* c, int n, const orc_agent* agents, const orc_lsc* lsc,
                    const long long* lsc_off, const orc_box* sfc, double tol, int max_iter, int threads,
                    double* x, double* obj, int* status, int* iters) {
    int nv = c->dim * c->M * (c->n + 1);
    int bad = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(+ : bad)
#endif
    for (int qi = 0; qi < n; qi++) {
        int it = 0;
        int st = orc_solve(c, &agents[qi], lsc ? &lsc[lsc_off[qi]] : NULL, sfc ? &sfc[(long long)qi * c->M] : NULL, tol,
                           max_iter, &x[(long long)qi * nv], &obj[qi], NULL, NULL, NULL, NULL, &it);
        status[qi] = st;
        if (iters) iters[qi] = it;
        if (st != 0) bad++;
    }
    return bad;
}
