/*
 * rsb_oracle.c — CPU fp64 oracle (see rsb_oracle.h: TEST INFRASTRUCTURE, PARITY UNPINNED).
 *
 * No reference file exists to cite; each block cites the published algorithm it
 * follows instead.  "RBDA" = Featherstone, Rigid Body Dynamics Algorithms, Springer 2008.
 *
 * Formulation.  All spatial quantities of one step are expressed in ONE frame: world-aligned
 * axes with origin O at the floating base's position at the start of the step.  In a common
 * frame the parent<->child Pluecker transforms of RBDA's recursions are identities, so the
 * recursions reduce to sums along the tree; and keeping O on the robot keeps |r| <= ~1 m so the
 * device's fp32 version of the same formulation does not lose digits when the robot walks away
 * from the world origin.  Spatial vectors are [angular(3); linear(3)].
 *
 * Generalized velocity follows RaiSim [RECALL]: u = [v_base (world), w_base (world), qdot].
 * The base therefore has S = identity and a velocity-product acceleration [0; -w x v]
 * (d/dt of the O-referenced linear velocity of a body whose origin moves with v).
 */
#include "rsb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXB RSB_MAX_BODIES
#define MAXV RSB_MAX_DOF
#define MAXK RSB_MAX_CONTACTS
#define ORC_WARM 6   /* warm state per collision primitive: impulse (3, contact frame), friction direction (2), direction valid */
#define ORC_SELF_REG 1e-4     /* compliance of a self-collision's Delassus block, relative to its mean diagonal (see step_impl) */
#define ORC_LAMBDA_FLOOR 1e-3 /* N s; keeps the relative convergence test meaningful as impulses -> 0 */

/* ------------------------------------------------------------------ small linear algebra */
static void cross3(const double* a, const double* b, double* c) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dot6(const double* a, const double* b) { return dot3(a, b) + dot3(a + 3, b + 3); }
static void mat3_mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof T);
}
static void mat3_vec(const double* A, const double* x, double* y) {
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
  y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
static void quat_to_rot(const double* qin, double* R) {
  double n = sqrt(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
  double w = qin[0] / n, x = qin[1] / n, y = qin[2] / n, z = qin[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
/* Rodrigues: R = I + s K + (1-c) K^2, K = [a]x, |a| = 1 */
static void axis_angle_rot(const double* a, double q, double* R) {
  double s = sin(q), c = cos(q), v = 1 - c;
  R[0] = c + a[0] * a[0] * v;        R[1] = a[0] * a[1] * v - a[2] * s; R[2] = a[0] * a[2] * v + a[1] * s;
  R[3] = a[1] * a[0] * v + a[2] * s; R[4] = c + a[1] * a[1] * v;        R[5] = a[1] * a[2] * v - a[0] * s;
  R[6] = a[2] * a[0] * v - a[1] * s; R[7] = a[2] * a[1] * v + a[0] * s; R[8] = c + a[2] * a[2] * v;
}
/* motion cross product  [w;v] x [a;s] = [w x a; w x s + v x a]   (RBDA eq. 2.31) */
static void crm(const double* V, const double* S, double* out) {
  double t1[3], t2[3], t3[3];
  cross3(V, S, t1); cross3(V, S + 3, t2); cross3(V + 3, S, t3);
  for (int i = 0; i < 3; ++i) { out[i] = t1[i]; out[3 + i] = t2[i] + t3[i]; }
}
/* force cross product  [w;v] x* [n;f] = [w x n + v x f; w x f]    (RBDA eq. 2.32) */
static void crf(const double* V, const double* F, double* out) {
  double t1[3], t2[3], t3[3];
  cross3(V, F, t1); cross3(V + 3, F + 3, t2); cross3(V, F + 3, t3);
  for (int i = 0; i < 3; ++i) { out[i] = t1[i] + t2[i]; out[3 + i] = t3[i]; }
}
static void mat6_vec(const double* A, const double* x, double* y) {
  double t[6];
  for (int i = 0; i < 6; ++i) { t[i] = 0; for (int j = 0; j < 6; ++j) t[i] += A[6 * i + j] * x[j]; }
  memcpy(y, t, sizeof t);
}
/* dense SPD solve (Cholesky), n <= 6, A row-major, in place on copies */
static void spd_solve(const double* Ain, const double* b, double* x, int n) {
  double L[36], y[6];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = Ain[n * i + j];
      for (int k = 0; k < j; ++k) s -= L[n * i + k] * L[n * j + k];
      L[n * i + j] = (i == j) ? sqrt(s) : s / L[n * j + j];
    }
  for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[n * i + k] * y[k]; y[i] = s / L[n * i + i]; }
  for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[n * k + i] * x[k]; x[i] = s / L[n * i + i]; }
}
static void inv3(const double* A, double* B) {
  double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
  double id = 1.0 / (A[0] * c0 + A[1] * c1 + A[2] * c2);
  B[0] = c0 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c1 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c2 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* inverse of a 6 x 6 matrix by Gauss-Jordan with partial pivoting; returns 0 when a pivot is below tol * (largest |entry|) */
static int inv6(const double* A, double* B, double tol) {
  double a[6][12], big = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) { a[i][j] = A[6 * i + j]; a[i][6 + j] = i == j ? 1.0 : 0.0; if (fabs(a[i][j]) > big) big = fabs(a[i][j]); }
  if (!(big > 0)) return 0;
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (fabs(a[piv][c]) < tol * big) return 0;
    if (piv != c) for (int j = 0; j < 12; ++j) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
    const double ip = 1.0 / a[c][c];
    for (int j = 0; j < 12; ++j) a[c][j] *= ip;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      if (f != 0.0) for (int j = 0; j < 12; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) B[6 * i + j] = a[i][6 + j];
  return 1;
}

/* --------------------------------------------------------------------------- kinematics */
typedef struct kin_t {
  double pbase[3];
  double R[MAXB][9], r[MAXB][3], a[MAXB][3], S[MAXB][6], V[MAXB][6];
  double com[MAXB][3];   /* relative to O */
  double Isp[MAXB][36];  /* spatial inertia about O, [[A,B],[B^T,m1]] */
} kin_t;

static int dof_of(int body) { return body + 5; }   /* body >= 1 */
static int qidx_of(int body) { return body + 6; }

/* RBDA §4.1 (model), eq. 2.63 (spatial inertia about a displaced origin) */
static void spatial_inertia(double mass, const double* c, const double* Ic, double* I6) {
  double cc = dot3(c, c);
  double A[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[3 * i + j] = Ic[3 * i + j] + mass * ((i == j ? cc : 0.0) - c[i] * c[j]);
  double B[9] = {0, -mass * c[2], mass * c[1], mass * c[2], 0, -mass * c[0], -mass * c[1], mass * c[0], 0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      I6[6 * i + j] = A[3 * i + j];
      I6[6 * i + 3 + j] = B[3 * i + j];
      I6[6 * (3 + i) + j] = B[3 * j + i];
      I6[6 * (3 + i) + 3 + j] = (i == j) ? mass : 0.0;
    }
}

static void kinematics(const rsb_model_blob* m, const double* q, const double* u, kin_t* k) {
  for (int i = 0; i < 3; ++i) { k->pbase[i] = q[i]; k->r[0][i] = 0; k->a[0][i] = 0; }
  quat_to_rot(q + 3, k->R[0]);
  for (int i = 0; i < 6; ++i) k->S[0][i] = 0;
  if (u) { for (int i = 0; i < 3; ++i) { k->V[0][i] = u[3 + i]; k->V[0][3 + i] = u[i]; } }
  else memset(k->V[0], 0, sizeof k->V[0]);
  for (int i = 1; i < m->nb; ++i) {
    int p = m->parent[i];
    double t[3];
    mat3_vec(k->R[p], m->ptree[i], t);
    for (int c = 0; c < 3; ++c) k->r[i][c] = k->r[p][c] + t[c];
    if (m->jtype[i] == RSB_JOINT_REVOLUTE) {
      double Rq[9], E[9];
      axis_angle_rot(m->axis[i], q[qidx_of(i)], Rq);
      mat3_mul(m->rtree[i], Rq, E);
      mat3_mul(k->R[p], E, k->R[i]);
      mat3_vec(k->R[i], m->axis[i], k->a[i]);
      for (int c = 0; c < 3; ++c) k->S[i][c] = k->a[i][c];
      cross3(k->r[i], k->a[i], k->S[i] + 3);
    } else { /* prismatic */
      mat3_mul(k->R[p], m->rtree[i], k->R[i]);
      mat3_vec(k->R[i], m->axis[i], k->a[i]);
      for (int c = 0; c < 3; ++c) { k->r[i][c] += k->a[i][c] * q[qidx_of(i)]; k->S[i][c] = 0; k->S[i][3 + c] = k->a[i][c]; }
    }
    double qd = u ? u[dof_of(i)] : 0.0;
    for (int c = 0; c < 6; ++c) k->V[i][c] = k->V[p][c] + k->S[i][c] * qd;
  }
  for (int i = 0; i < m->nb; ++i) {
    double t[3], Il[9], T[9], Iw[9], Rt[9];
    mat3_vec(k->R[i], m->com[i], t);
    for (int c = 0; c < 3; ++c) k->com[i][c] = k->r[i][c] + t[c];
    const double* I = m->inertia[i];
    Il[0] = I[0]; Il[1] = I[1]; Il[2] = I[2]; Il[3] = I[1]; Il[4] = I[3]; Il[5] = I[4]; Il[6] = I[2]; Il[7] = I[4]; Il[8] = I[5];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Rt[3 * a + b] = k->R[i][3 * b + a];
    mat3_mul(k->R[i], Il, T);
    mat3_mul(T, Rt, Iw);
    spatial_inertia(m->mass[i], k->com[i], Iw, k->Isp[i]);
  }
}

/* RNEA in the common frame (RBDA Table 5.1 with identity transforms). udot may be NULL. */
static void rnea(const rsb_model_blob* m, const kin_t* k, const double* u, const double* udot,
                 const double* gravity, double* tau) {
  double A[MAXB][6], F[MAXB][6];
  double wxv[3];
  cross3(k->V[0], k->V[0] + 3, wxv);
  for (int c = 0; c < 3; ++c) {
    A[0][c] = udot ? udot[3 + c] : 0.0;
    A[0][3 + c] = (udot ? udot[c] : 0.0) - wxv[c] - gravity[c];
  }
  for (int i = 1; i < m->nb; ++i) {
    int p = m->parent[i];
    double vs[6];
    crm(k->V[p], k->S[i], vs);
    double qd = u ? u[dof_of(i)] : 0.0, qdd = udot ? udot[dof_of(i)] : 0.0;
    for (int c = 0; c < 6; ++c) A[i][c] = A[p][c] + k->S[i][c] * qdd + vs[c] * qd;
  }
  for (int i = 0; i < m->nb; ++i) {
    double IA[6], IV[6], vf[6];
    mat6_vec(k->Isp[i], A[i], IA);
    mat6_vec(k->Isp[i], k->V[i], IV);
    crf(k->V[i], IV, vf);
    for (int c = 0; c < 6; ++c) F[i][c] = IA[c] + vf[c];
  }
  for (int i = m->nb - 1; i >= 1; --i) {
    int p = m->parent[i];
    tau[dof_of(i)] = dot6(k->S[i], F[i]);
    for (int c = 0; c < 6; ++c) F[p][c] += F[i][c];
  }
  for (int c = 0; c < 3; ++c) { tau[c] = F[0][3 + c]; tau[3 + c] = F[0][c]; }
}

/* CRBA in the common frame (RBDA Table 6.2 with identity transforms) */
static void crba(const rsb_model_blob* m, const kin_t* k, double* M) {
  int nv = m->nv;
  static const int perm[6] = {3, 4, 5, 0, 1, 2}; /* gv index -> spatial index */
  double Ic[MAXB][36];
  memcpy(Ic, k->Isp, sizeof(double) * 36 * m->nb);
  memset(M, 0, sizeof(double) * nv * nv);
  for (int i = m->nb - 1; i >= 1; --i) {
    int p = m->parent[i];
    for (int c = 0; c < 36; ++c) Ic[p][c] += Ic[i][c];
  }
  for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) M[a * nv + b] = Ic[0][6 * perm[a] + perm[b]];
  for (int i = 1; i < m->nb; ++i) {
    double F[6];
    mat6_vec(Ic[i], k->S[i], F);
    int di = dof_of(i);
    M[di * nv + di] = dot6(k->S[i], F) + m->armature[i];
    for (int j = m->parent[i]; j >= 1; j = m->parent[j]) {
      int dj = dof_of(j);
      M[di * nv + dj] = M[dj * nv + di] = dot6(k->S[j], F);
    }
    for (int a = 0; a < 6; ++a) M[di * nv + a] = M[a * nv + di] = F[perm[a]];
  }
}

/* dof-level parent array ("lambda" of RBDA §6.5): base dofs form a chain 0<-1<-...<-5 */
static void dof_parents(const rsb_model_blob* m, int* pd) {
  pd[0] = -1;
  for (int k = 1; k < 6; ++k) pd[k] = k - 1;
  for (int i = 1; i < m->nb; ++i) pd[dof_of(i)] = m->parent[i] == 0 ? 5 : dof_of(m->parent[i]);
}

/* LTDL factorisation exploiting branch-induced sparsity, in place (RBDA Table 6.3).
 * On exit: diagonal = D, strictly-lower entries on ancestor positions = L (unit diagonal). */
static void ltdl(double* H, int nv, const int* pd) {
  for (int k = nv - 1; k >= 0; --k) {
    int i = pd[k];
    while (i >= 0) {
      double a = H[k * nv + i] / H[k * nv + k];
      int j = i;
      while (j >= 0) { H[i * nv + j] -= a * H[k * nv + j]; j = pd[j]; }
      H[k * nv + i] = a;
      i = pd[i];
    }
  }
}
/* x <- M^-1 x given the LTDL factors (RBDA Table 6.5) */
static void ltdl_solve(const double* H, int nv, const int* pd, double* x) {
  for (int k = nv - 1; k >= 0; --k) for (int i = pd[k]; i >= 0; i = pd[i]) x[i] -= H[k * nv + i] * x[k];
  for (int k = 0; k < nv; ++k) x[k] /= H[k * nv + k];
  for (int k = 0; k < nv; ++k) for (int i = pd[k]; i >= 0; i = pd[i]) x[k] -= H[k * nv + i] * x[i];
}

/* 3 x nv Jacobian (world axes) of a point x (relative to O) fixed on `body` */
static void point_jacobian(const rsb_model_blob* m, const kin_t* k, int body, const double* x, double* J) {
  int nv = m->nv;
  memset(J, 0, sizeof(double) * 3 * nv);
  for (int c = 0; c < 3; ++c) J[c * nv + c] = 1.0;
  /* v = v_b + w x x  =>  d/dw = -[x]x */
  J[0 * nv + 4] = x[2];  J[0 * nv + 5] = -x[1];
  J[1 * nv + 3] = -x[2]; J[1 * nv + 5] = x[0];
  J[2 * nv + 3] = x[1];  J[2 * nv + 4] = -x[0];
  for (int j = body; j >= 1; j = m->parent[j]) {
    double col[3];
    if (m->jtype[j] == RSB_JOINT_REVOLUTE) {
      double d[3] = {x[0] - k->r[j][0], x[1] - k->r[j][1], x[2] - k->r[j][2]};
      cross3(k->a[j], d, col);
    } else { col[0] = k->a[j][0]; col[1] = k->a[j][1]; col[2] = k->a[j][2]; }
    for (int c = 0; c < 3; ++c) J[c * nv + dof_of(j)] = col[c];
  }
}

/* ------------------------------------------------------------------------- public queries */
void orc_default_params(orc_params* p) {
  memset(p, 0, sizeof *p);
  p->dt = 0.0025;
  p->gravity[2] = -9.81;
  p->mu = 0.8;
  p->erp = 0.0;
  p->alpha_init = 1.0; p->alpha_min = 1.0; p->alpha_decay = 1.0;
  p->threshold = 1e-5;
  p->max_iter = 150;
  p->section_rounds = 2;
  p->freeze_after = 6;   /* sweeps before a slipping contact keeps its direction.  Benchmark population vs the plain iteration (tests/test_oracle_solver_heuristics.py):
                            freeze_after 10: |du| p99.9 2.2e-6 m/s, 5 solves > 1e-4 | 8: 2.3e-6, 8 | 6: 3.4e-6, 9 | 5: 7.4e-6, 12 | 4: 2.6e-5, 16 (fails the 1e-5 bound);
                            device throughput (config 2): 130 / 136 / 143 / 148 M env-steps/s at 10 / 8 / 6 / 5 */
  p->refine = 1;
  p->group_parallel = 1; /* what the device runs: grouped sweep (block Jacobi across limbs, Gauss-Seidel within a limb) */
  p->dir_per_sweep = 1;  /* only with group_parallel = 0 (ablations): 1 = sequential sweep with one direction refresh per sweep (the device
                            until round 2b), 0 = the round-1 scheme (a refinement inside every contact update) */
  p->settle_tol = 0.0;   /* off: freezing a direction whose last refinement moved it by < 1e-4 rad saved 19 % of the Newton refinements and
                            no sweeps, but put the p99.9 velocity deviation from the plain per-contact iteration at 1.9e-4 m/s instead of
                            7e-6 (tests/test_oracle_solver_heuristics.py) */
  p->restitution = 0.0; p->res_threshold = 0.0;
  p->self_collision = 1;   /* RaiSim's default: the links of one system collide with each other (parent-child pairs excepted) */
  p->warm_start = 1;  /* only has an effect when the caller carries a warm state (orc_step_warm / orc_step_batch with lam_warm):
                         8% fewer sweeps and 7x fewer global searches on the config-2 workload.  The device keeps the same state
                         per env (StepArgs::warm, rsb_set_solver_warm_start, default on), so parity tests over several
                         integrate() calls and the CPU baseline carry a warm state too. */
  p->stall_window = 4;
  p->stall_factor = 0.5;
  p->kmax = 8;
  p->control_mode = RSB_PD_PLUS_FEEDFORWARD_TORQUE;
  p->terrain_type = 0;
  p->ground_z = 0.0;
  p->hm_index = NULL;
  p->multi_depth = 3; p->multi_light = 0; p->multi_freeze_after = 0; p->multi_stall_window = 16;
  p->anderson = 2; p->anderson_clip = 20.0;
  p->hm_contacts = 1; p->hm_second_cos = 0.70710678118654752;
  p->hm_capsule = 0;
  p->integ_theta = 1.0;
}

void orc_mass_matrix(const rsb_model_blob* m, const double* q, double* M) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  kinematics(m, q, NULL, k);
  crba(m, k, M);
  free(k);
}

void orc_mass_matrix_rne(const rsb_model_blob* m, const double* q, double* M) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  kinematics(m, q, NULL, k);
  double zero[3] = {0, 0, 0}, e[MAXV], col[MAXV];
  for (int j = 0; j < m->nv; ++j) {
    memset(e, 0, sizeof e);
    e[j] = 1.0;
    rnea(m, k, NULL, e, zero, col);
    for (int i = 0; i < m->nv; ++i) M[i * m->nv + j] = col[i] + ((i == j && i >= 6) ? m->armature[i - 5] : 0.0);
  }
  free(k);
}

void orc_nonlinearities(const rsb_model_blob* m, const orc_params* p, const double* q, const double* u, double* h) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  kinematics(m, q, u, k);
  rnea(m, k, u, NULL, p->gravity, h);
  free(k);
}

void orc_inverse_dynamics(const rsb_model_blob* m, const orc_params* p, const double* q, const double* u,
                          const double* udot, double* tau) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  kinematics(m, q, u, k);
  rnea(m, k, u, udot, p->gravity, tau);
  for (int i = 1; i < m->nb; ++i) tau[dof_of(i)] += m->armature[i] * udot[dof_of(i)];
  free(k);
}

void orc_forward_dynamics(const rsb_model_blob* m, const orc_params* p, const double* q, const double* u,
                          const double* tau, double* udot) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  int nv = m->nv, pd[MAXV];
  double* M = (double*)malloc(sizeof(double) * nv * nv);
  double h[MAXV];
  kinematics(m, q, u, k);
  crba(m, k, M);
  rnea(m, k, u, NULL, p->gravity, h);
  dof_parents(m, pd);
  ltdl(M, nv, pd);
  for (int i = 0; i < nv; ++i) udot[i] = tau[i] - h[i];
  ltdl_solve(M, nv, pd, udot);
  free(M); free(k);
}

/* Articulated-body algorithm in the common frame (RBDA Table 7.1 with identity transforms). */
void orc_aba(const rsb_model_blob* m, const orc_params* p, const double* q, const double* u,
             const double* tau, double* udot) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  kinematics(m, q, u, k);
  int nb = m->nb;
  double (*IA)[36] = (double (*)[36])malloc(sizeof(double) * 36 * nb);
  double pA[MAXB][6], cb[MAXB][6], U[MAXB][6], D[MAXB], uu[MAXB], A[MAXB][6];
  for (int i = 0; i < nb; ++i) {
    double IV[6];
    memcpy(IA[i], k->Isp[i], sizeof(double) * 36);
    mat6_vec(k->Isp[i], k->V[i], IV);
    crf(k->V[i], IV, pA[i]);
    if (i >= 1) {
      double vs[6];
      crm(k->V[m->parent[i]], k->S[i], vs);
      for (int c = 0; c < 6; ++c) cb[i][c] = vs[c] * u[dof_of(i)];
    }
  }
  for (int i = nb - 1; i >= 1; --i) {
    int par = m->parent[i];
    mat6_vec(IA[i], k->S[i], U[i]);
    D[i] = dot6(k->S[i], U[i]) + m->armature[i];
    uu[i] = tau[dof_of(i)] - dot6(k->S[i], pA[i]);
    double Ia[36], Iac[6];
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) Ia[6 * a + b] = IA[i][6 * a + b] - U[i][a] * U[i][b] / D[i];
    mat6_vec(Ia, cb[i], Iac);
    for (int c = 0; c < 36; ++c) IA[par][c] += Ia[c];
    for (int c = 0; c < 6; ++c) pA[par][c] += pA[i][c] + Iac[c] + U[i][c] * uu[i] / D[i];
  }
  /* base: IA_0 A_0 + pA_0 = [tau_ang; tau_lin] */
  double rhs[6], A0[6], wxv[3];
  for (int c = 0; c < 3; ++c) { rhs[c] = tau[3 + c] - pA[0][c]; rhs[3 + c] = tau[c] - pA[0][3 + c]; }
  spd_solve(IA[0], rhs, A0, 6);
  memcpy(A[0], A0, sizeof A0);
  cross3(k->V[0], k->V[0] + 3, wxv);
  for (int c = 0; c < 3; ++c) { udot[3 + c] = A0[c]; udot[c] = A0[3 + c] + wxv[c] + p->gravity[c]; }
  for (int i = 1; i < nb; ++i) {
    int par = m->parent[i];
    double Ap[6];
    for (int c = 0; c < 6; ++c) Ap[c] = A[par][c] + cb[i][c];
    double qdd = (uu[i] - dot6(U[i], Ap)) / D[i];
    udot[dof_of(i)] = qdd;
    for (int c = 0; c < 6; ++c) A[i][c] = Ap[c] + k->S[i][c] * qdd;
  }
  free(IA); free(k);
}

void orc_point_jacobian(const rsb_model_blob* m, const double* q, int body, const double* p_local,
                        double* pos_world, double* J) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  kinematics(m, q, NULL, k);
  double t[3], x[3];
  mat3_vec(k->R[body], p_local, t);
  for (int c = 0; c < 3; ++c) { x[c] = k->r[body][c] + t[c]; pos_world[c] = k->pbase[c] + x[c]; }
  point_jacobian(m, k, body, x, J);
  free(k);
}

void orc_energy(const rsb_model_blob* m, const orc_params* p, const double* q, const double* u,
                double* kinetic, double* potential) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  kinematics(m, q, u, k);
  double T = 0, Upot = 0;
  for (int i = 0; i < m->nb; ++i) {
    double IV[6];
    mat6_vec(k->Isp[i], k->V[i], IV);
    T += 0.5 * dot6(k->V[i], IV);
    if (i >= 1) T += 0.5 * m->armature[i] * u[dof_of(i)] * u[dof_of(i)];
    double cw[3] = {k->pbase[0] + k->com[i][0], k->pbase[1] + k->com[i][1], k->pbase[2] + k->com[i][2]};
    Upot -= m->mass[i] * dot3(p->gravity, cw);
  }
  *kinetic = T; *potential = Upot;
  free(k);
}

void orc_momentum(const rsb_model_blob* m, const double* q, const double* u, double* lin, double* ang) {
  kin_t* k = (kin_t*)malloc(sizeof(kin_t));
  kinematics(m, q, u, k);
  double H[6] = {0};
  for (int i = 0; i < m->nb; ++i) {
    double IV[6];
    mat6_vec(k->Isp[i], k->V[i], IV);
    for (int c = 0; c < 6; ++c) H[c] += IV[c];
  }
  /* shift the angular momentum from O to the world origin: L_0 = L_O + p_base x P */
  double pxP[3];
  cross3(k->pbase, H + 3, pxP);
  for (int c = 0; c < 3; ++c) { lin[c] = H[3 + c]; ang[c] = H[c] + pxP[c]; }
  free(k);
}

long orc_pair_evals = 0;   /* EXPERIMENT orc_params::pair_inner: one-contact rule evaluations spent inside joint pair solves (not thread safe: statistics of single-threaded runs) */

/* ------------------------------------------------------------------------------ terrain */
void orc_terrain(const orc_params* p, double x, double y, double* h, double* n) {
  if (p->terrain_type == 0) { *h = p->ground_z; n[0] = 0; n[1] = 0; n[2] = 1; return; }
  /* regular grid, each cell split along the (ix,iy)-(ix+1,iy+1) diagonal into two triangles */
  int xs = p->hm_xs, ys = p->hm_ys;
  double dx = p->hm_xsize / (xs - 1), dy = p->hm_ysize / (ys - 1);
  double gx = (x - (p->hm_cx - 0.5 * p->hm_xsize)) / dx, gy = (y - (p->hm_cy - 0.5 * p->hm_ysize)) / dy;
  if (gx < 0) gx = 0;
  if (gx > xs - 1) gx = xs - 1;
  if (gy < 0) gy = 0;
  if (gy > ys - 1) gy = ys - 1;
  int ix = (int)floor(gx), iy = (int)floor(gy);
  if (ix > xs - 2) ix = xs - 2;
  if (iy > ys - 2) iy = ys - 2;
  double fx = gx - ix, fy = gy - iy;
  double h00 = p->hm_heights[iy * xs + ix], h10 = p->hm_heights[iy * xs + ix + 1];
  double h01 = p->hm_heights[(iy + 1) * xs + ix], h11 = p->hm_heights[(iy + 1) * xs + ix + 1];
  double sx, sy;
  if (fx >= fy) { sx = h10 - h00; sy = h11 - h10; } else { sx = h11 - h01; sy = h01 - h00; }
  *h = h00 + sx * fx + sy * fy;
  double gxs = sx / dx, gys = sy / dy, inv = 1.0 / sqrt(gxs * gxs + gys * gys + 1.0);
  n[0] = -gxs * inv; n[1] = -gys * inv; n[2] = inv;
}

/* Closest point of the triangle (a, b, c) to the ORIGIN (the caller shifts the sphere centre there): the Voronoi-region
 * walk of Ericson, "Real-Time Collision Detection" (2005), section 5.1.5 - vertex, edge and face regions. */
static void closest_on_triangle(const double* a, const double* b, const double* c, double* out) {
  double ab[3], ac[3], bc[3];
  for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; bc[i] = c[i] - b[i]; }
  const double d1 = -dot3(ab, a), d2 = -dot3(ac, a);
  const double d3 = -dot3(ab, b), d4 = -dot3(ac, b);
  const double d5 = -dot3(ab, c), d6 = -dot3(ac, c);
  const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  double base[3] = {a[0], a[1], a[2]}, dir1[3] = {0, 0, 0}, dir2[3] = {0, 0, 0}, t1 = 0.0, t2 = 0.0;
  if (d1 <= 0.0 && d2 <= 0.0) { /* vertex a */ }
  else if (d3 >= 0.0 && d4 <= d3) { for (int i = 0; i < 3; ++i) base[i] = b[i]; }
  else if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { t1 = d1 / (d1 - d3); for (int i = 0; i < 3; ++i) dir1[i] = ab[i]; }
  else if (d6 >= 0.0 && d5 <= d6) { for (int i = 0; i < 3; ++i) base[i] = c[i]; }
  else if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { t1 = d2 / (d2 - d6); for (int i = 0; i < 3; ++i) dir1[i] = ac[i]; }
  else if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    t1 = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int i = 0; i < 3; ++i) { base[i] = b[i]; dir1[i] = bc[i]; }
  } else {
    const double den = 1.0 / (va + vb + vc);
    t1 = vb * den; t2 = vc * den;
    for (int i = 0; i < 3; ++i) { dir1[i] = ab[i]; dir2[i] = ac[i]; }
  }
  for (int i = 0; i < 3; ++i) out[i] = base[i] + t1 * dir1[i] + t2 * dir2[i];
}

/* Narrow phase sphere (centre c, world frame; radius r) x terrain: penetration depth and unit contact normal.
 * Plane: depth = r - (c_z - z0), normal = z.  Height map: the CLOSEST FEATURE (face, edge or vertex) of the triangulated
 * surface over the cells the sphere's xy bounding square overlaps (at most ORC_HM_CELLS x ORC_HM_CELLS of them, scanned
 * row by row, the lower-right triangle of a cell first; the first of equally close features wins): normal = from the
 * closest point to the centre, depth = r - distance.  A centre at or below the surface, or beyond the map's border, falls
 * back to the plane of the triangle under it (see the end of the function).
 * Upstream counterpart: the sphere x HeightMap collider of RaiSim's vendored ODE - absent from /root/reference. */
#define ORC_HM_CELLS 3
/* depth2 / n2 (may be NULL): with orc_params::hm_contacts >= 2 a sphere in a valley also reports the closest feature of a SECOND flank -
 * the closest penetrating triangle point whose direction differs from the first contact's normal by more than acos(hm_second_cos);
 * *depth2 <= 0 when there is none. */
static int terrain_contact_ex(const orc_params* p, const double* c, double r, double* depth, double* n, double* depth2, double* n2, int above_test);
static int terrain_contact(const orc_params* p, const double* c, double r, double* depth, double* n, double* depth2, double* n2) {
  return terrain_contact_ex(p, c, r, depth, n, depth2, n2, p->hm_plane_test ? 0 : 1);     /* round 5: the height field decides (orc_params::hm_plane_test) */
}
/* above_test = 1 (since round 5 every caller; round 4: the capsule search only): "the centre is outside the terrain" is decided by the height field
 * itself (c_z above the surface at (c_x, c_y)) instead of by the plane of the triangle that holds the closest point.  At a CONVEX edge sharper
 * than the sphere is close the two differ: past the ridge line the centre is below the extended plane of the first face although it is above the
 * surface, and the plane test then falls back to the plane of the face under the centre - depth and normal of the wrong feature (a sphere resting
 * against a kerb; for the capsule search it ranked a point beside the ridge deeper than the point above it).  above_test = 0 survives as
 * orc_params::hm_plane_test for the KAT that shows the difference. */
static int terrain_contact_ex(const orc_params* p, const double* c, double r, double* depth, double* n, double* depth2, double* n2, int above_test) {
  if (depth2) *depth2 = 0.0;
  if (p->terrain_type == 0) {
    n[0] = 0; n[1] = 0; n[2] = 1;
    *depth = r - (c[2] - p->ground_z);
    return *depth > 0.0;
  }
  const int xs = p->hm_xs, ys = p->hm_ys;
  const double dx = p->hm_xsize / (xs - 1), dy = p->hm_ysize / (ys - 1);
  const double x0 = p->hm_cx - 0.5 * p->hm_xsize, y0 = p->hm_cy - 0.5 * p->hm_ysize;
  int ix0 = (int)floor((c[0] - r - x0) / dx), ix1 = (int)floor((c[0] + r - x0) / dx);
  int iy0 = (int)floor((c[1] - r - y0) / dy), iy1 = (int)floor((c[1] + r - y0) / dy);
  const int icx = (int)floor((c[0] - x0) / dx), icy = (int)floor((c[1] - y0) / dy);
  if (ix1 - ix0 >= ORC_HM_CELLS) { ix0 = icx - ORC_HM_CELLS / 2; ix1 = ix0 + ORC_HM_CELLS - 1; }
  if (iy1 - iy0 >= ORC_HM_CELLS) { iy0 = icy - ORC_HM_CELLS / 2; iy1 = iy0 + ORC_HM_CELLS - 1; }
  if (ix0 < 0) ix0 = 0;
  if (iy0 < 0) iy0 = 0;
  if (ix1 > xs - 2) ix1 = xs - 2;
  if (iy1 > ys - 2) iy1 = ys - 2;
  if (ix0 > ix1) { ix0 = ix1 = ix0 > xs - 2 ? xs - 2 : 0; }     /* beyond the map's border: its outermost cells */
  if (iy0 > iy1) { iy0 = iy1 = iy0 > ys - 2 ? ys - 2 : 0; }
  double best = 1e300, bp[3] = {0, 0, 0}, bn[3] = {0, 0, 1};
  for (int iy = iy0; iy <= iy1; ++iy)
    for (int ix = ix0; ix <= ix1; ++ix) {
      const float* H = p->hm_heights + iy * xs + ix;
      const double ox = x0 + ix * dx - c[0], oy = y0 + iy * dy - c[1];
      const double v00[3] = {ox, oy, H[0] - c[2]}, v10[3] = {ox + dx, oy, H[1] - c[2]};
      const double v01[3] = {ox, oy + dy, H[xs] - c[2]}, v11[3] = {ox + dx, oy + dy, H[xs + 1] - c[2]};
      for (int tri = 0; tri < 2; ++tri) {
        const double* b = tri == 0 ? v10 : v11;
        const double* cc = tri == 0 ? v11 : v01;
        double q[3];
        closest_on_triangle(v00, b, cc, q);
        const double d2 = dot3(q, q);
        /* equally close features (a point on an edge two triangles share) are ranked by the scan order, not by rounding noise:
         * a later candidate must be closer by more than 4e-6 (relative, squared distance) to win - the device ranks its
         * fp32 candidates the same way (5 mantissa bits of the squared distance carry the scan position) */
        if (d2 < best * (1.0 - 4e-6)) {
          best = d2;
          double e1[3], e2[3];
          for (int i = 0; i < 3; ++i) { bp[i] = q[i]; e1[i] = b[i] - v00[i]; e2[i] = cc[i] - v00[i]; }
          cross3(e1, e2, bn);
        }
      }
    }
  const double dist = sqrt(best);
  const int inside = c[0] >= x0 && c[0] <= x0 + p->hm_xsize && c[1] >= y0 && c[1] <= y0 + p->hm_ysize;
  int outer = -dot3(bp, bn) > 0.0;
  if (above_test) { double hh, nh[3]; orc_terrain(p, c[0], c[1], &hh, nh); outer = c[2] > hh; }
  if (inside && outer && dist > 1e-9) {
    for (int i = 0; i < 3; ++i) n[i] = -bp[i] / dist;
    *depth = r - dist;
    if (depth2 && p->hm_contacts >= 2 && *depth > 0.0) {
      /* second flank: the same scan, restricted to points that penetrate, lie on the outer side of their triangle and whose direction
       * is at least acos(hm_second_cos) away from the first normal (coplanar neighbours and the far side of a shared edge give the
       * first contact's direction again and drop out) */
      double best2 = r * r, q2[3] = {0, 0, 0};
      int found = 0;
      for (int iy = iy0; iy <= iy1; ++iy)
        for (int ix = ix0; ix <= ix1; ++ix) {
          const float* H = p->hm_heights + iy * xs + ix;
          const double ox = x0 + ix * dx - c[0], oy = y0 + iy * dy - c[1];
          const double v00[3] = {ox, oy, H[0] - c[2]}, v10[3] = {ox + dx, oy, H[1] - c[2]};
          const double v01[3] = {ox, oy + dy, H[xs] - c[2]}, v11[3] = {ox + dx, oy + dy, H[xs + 1] - c[2]};
          for (int tri = 0; tri < 2; ++tri) {
            const double* b = tri == 0 ? v10 : v11;
            const double* cc = tri == 0 ? v11 : v01;
            double q[3], e1[3], e2[3], tn[3];
            closest_on_triangle(v00, b, cc, q);
            const double d2 = dot3(q, q);
            if (!(d2 < best2 * (1.0 - 4e-6)) || d2 < 1e-18) continue;
            for (int i = 0; i < 3; ++i) { e1[i] = b[i] - v00[i]; e2[i] = cc[i] - v00[i]; }
            cross3(e1, e2, tn);
            if (!(-dot3(q, tn) > 0.0)) continue;
            if (!(-dot3(q, n) < p->hm_second_cos * sqrt(d2))) continue;     /* direction . n1 < cos */
            best2 = d2; found = 1;
            for (int i = 0; i < 3; ++i) q2[i] = q[i];
          }
        }
      if (found) {
        const double dist2 = sqrt(best2);
        for (int i = 0; i < 3; ++i) n2[i] = -q2[i] / dist2;
        *depth2 = r - dist2;
      }
    }
  } else {
    /* the centre is at or below the surface (a zero-radius box corner, a sphere pushed in by more than its radius) or beyond
     * the map's border (the terrain continues flat from its outermost samples): the triangle UNDER the centre decides -
     * depth = r - distance to its plane, normal = its face normal */
    double h;
    orc_terrain(p, c[0], c[1], &h, n);
    *depth = r - (c[2] - h) * n[2];
  }
  return *depth > 0.0;
}

/* Capsule x height map (orc_params::hm_capsule).  A capsule is stored as its two end spheres (rsb_model_blob::col_capsule); on a plane
 * they are its exact contact set.  Against a height map the cylinder between them can touch where neither end does.  The deepest
 * point of the capsule's AXIS SEGMENT a + t (b - a), t in (0, 1), is located by a nested sampling of the sphere narrow phase above -
 * four rounds of four samples, each round centred on the best sample of the round before with 0.4 x its spacing (final spacing
 * 1.3 % of the capsule's length) - and reported as a contact of its own when it penetrates AND is deeper than both end spheres by
 * more than ORC_CAPSULE_MARGIN (a capsule lying on flat ground is held by its two ends: a third, redundant contact between them
 * would only slow the solver down).  Face, edge and vertex contacts all come from the same closest-feature test; the depth along
 * the segment is not unimodal over rough terrain, the first round's four samples decide which dip is refined.
 * A CYLINDER's barrel is the same search between its two cap centres (its ends are rim primitives: radius 0, rim = the cylinder's radius) with
 * t kept r / L away from either end: a sample sphere of the cylinder's radius then lies inside the cylinder (it reaches the flat cap, not past
 * it), and at an interior minimum of the distance the normal is perpendicular to the axis, where sphere and barrel coincide.
 * The device runs the same rounds (step_kernel.h, class-4 kernels): lane = (sample, cell).
 * Upstream counterpart: ODE's capsule x height-field collider - absent from /root/reference. */
#define ORC_CAPSULE_MARGIN 1e-4
#define ORC_CAPSULE_ROUNDS 4
static int capsule_contact(const orc_params* p, const double* a, const double* b, double r, double tmin, double dep_ends, double* c_out, double* depth, double* n) {
  static const double off[4] = {-0.6, -0.2, 0.2, 0.6};
  if (p->terrain_type != 1 || !(tmin < 0.5)) return 0;
  const double tmax = 1.0 - tmin;
  double c = 0.5, w = 0.5, best_d = 0.0, best_n[3] = {0, 0, 1}, best_t = 0.5;
  for (int round = 0; round < ORC_CAPSULE_ROUNDS; ++round) {
    int have = 0;
    for (int k = 0; k < 4; ++k) {
      double t = c + w * off[k];
      t = t < tmin ? tmin : (t > tmax ? tmax : t);
      const double pt[3] = {a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]), a[2] + t * (b[2] - a[2])};
      double d, nn[3];
      terrain_contact_ex(p, pt, r, &d, nn, NULL, NULL, 1);
      /* a later sample must be deeper by more than 2e-6 r to win: equal depths (a flat stretch) are ranked by the sample order on
       * both sides, not by rounding noise */
      if (!have || d > best_d + 2e-6 * r) { have = 1; best_d = d; best_t = t; for (int i = 0; i < 3; ++i) best_n[i] = nn[i]; }
    }
    c = best_t; w *= 0.4;
  }
  if (!(best_d > 0.0 && best_d > dep_ends + ORC_CAPSULE_MARGIN)) return 0;
  for (int i = 0; i < 3; ++i) { c_out[i] = a[i] + best_t * (b[i] - a[i]); n[i] = best_n[i]; }
  *depth = best_d;
  return 1;
}

/* Box x height map (orc_params::hm_capsule as well).  A box is stored as its eight corners (exact on a plane).  Against a height map a FACE or an
 * EDGE can touch where no corner does (a slab lying on a bump, a beam across a ridge).  The surface is piecewise linear and the box convex, so the
 * vertical penetration h(x, y) - z_box(x, y) of the box's lower surface takes its maximum at one of finitely many candidates:
 *   (corners)   the eight corner primitives themselves;
 *   (A)         a terrain VERTEX under the box: the vertical line through it enters the box at z_lo = the largest of the three slab entries
 *               (box = intersection of three slabs |(p - centre) . a_k| <= l_k), through the face of that slab;  normal = that face's inward normal;
 *   (B)         a CROSSING (in plan view) of one of the twelve box edges with a terrain edge - the grid lines x = const, y = const and the cells'
 *               diagonals;  normal = box edge x terrain edge, pointing up.
 * Depth of a candidate = vertical penetration x n_z.  All candidates within ORC_BOX_TIE of the deepest form the contact patch (a face lying flat on a
 * plateau: every plateau vertex); the ONE contact reported is their mean position and mean normal, when that deepest candidate penetrates and is
 * deeper than every corner by more than ORC_CAPSULE_MARGIN.  Two passes over the candidates (the maximum, then the patch): the result does not
 * depend on the order of enumeration (device: lane = candidate). */
#define ORC_BOX_TIE 1e-5
#define ORC_BOX_SPAN 32            /* terrain vertices examined per axis under one box, crossings per edge and family: beyond, the contact-overflow flag */
typedef struct { double dmax; double sum[7]; int pass; } box_acc;
static void box_candidate(box_acc* A, double d, const double* pos, const double* n) {
  if (A->pass == 0) { if (d > A->dmax) A->dmax = d; return; }
  if (d >= A->dmax - ORC_BOX_TIE) { A->sum[0] += 1.0; for (int a = 0; a < 3; ++a) { A->sum[1 + a] += pos[a]; A->sum[4 + a] += n[a]; } }
}
static int box_face_contact(const orc_params* p, const double cw[8][3], double dep_corners, double* c_out, double* depth, double* n_out, int* overflow) {
  if (p->terrain_type != 1) return 0;
  const int xs = p->hm_xs, ys = p->hm_ys;
  const double dx = p->hm_xsize / (xs - 1), dy = p->hm_ysize / (ys - 1), x0 = p->hm_cx - 0.5 * p->hm_xsize, y0 = p->hm_cy - 0.5 * p->hm_ysize;
  double ctr[3], e[3][3], ax[3][3], len[3];
  for (int a = 0; a < 3; ++a) {
    ctr[a] = 0.5 * (cw[0][a] + cw[7][a]);
    e[0][a] = 0.5 * (cw[1][a] - cw[0][a]); e[1][a] = 0.5 * (cw[2][a] - cw[0][a]); e[2][a] = 0.5 * (cw[4][a] - cw[0][a]);
  }
  for (int k = 0; k < 3; ++k) {
    len[k] = sqrt(dot3(e[k], e[k]));
    if (!(len[k] > 0.0)) return 0;
    for (int a = 0; a < 3; ++a) ax[k][a] = e[k][a] / len[k];
  }
  const double X = fabs(e[0][0]) + fabs(e[1][0]) + fabs(e[2][0]), Y = fabs(e[0][1]) + fabs(e[1][1]) + fabs(e[2][1]);
  int ix_lo = (int)ceil((ctr[0] - X - x0) / dx), ix_hi = (int)floor((ctr[0] + X - x0) / dx);
  int iy_lo = (int)ceil((ctr[1] - Y - y0) / dy), iy_hi = (int)floor((ctr[1] + Y - y0) / dy);
  if (ix_lo < 0) ix_lo = 0;
  if (iy_lo < 0) iy_lo = 0;
  if (ix_hi > xs - 1) ix_hi = xs - 1;
  if (iy_hi > ys - 1) iy_hi = ys - 1;
  if (ix_hi - ix_lo + 1 > ORC_BOX_SPAN) { ix_hi = ix_lo + ORC_BOX_SPAN - 1; *overflow = 1; }
  if (iy_hi - iy_lo + 1 > ORC_BOX_SPAN) { iy_hi = iy_lo + ORC_BOX_SPAN - 1; *overflow = 1; }
  box_acc A;
  A.dmax = -1e300;
  for (int a = 0; a < 7; ++a) A.sum[a] = 0.0;
  for (A.pass = 0; A.pass < 2; ++A.pass) {
    /* (A) terrain vertices */
    for (int iy = iy_lo; iy <= iy_hi; ++iy)
      for (int ix = ix_lo; ix <= ix_hi; ++ix) {
        const double x = x0 + ix * dx, y = y0 + iy * dy, h = p->hm_heights[iy * xs + ix];
        const double rx = x - ctr[0], ry = y - ctr[1];
        double zlo = -1e300, zup = 1e300, nz = 1.0;
        int face = 0, inside = 1;
        for (int k = 0; k < 3; ++k) {
          const double rho = rx * ax[k][0] + ry * ax[k][1], az = ax[k][2];
          if (fabs(az) < 1e-6) { if (fabs(rho) > len[k]) inside = 0; continue; }
          const double za = ctr[2] + (-len[k] - rho) / az, zb = ctr[2] + (len[k] - rho) / az;
          const double lo = za < zb ? za : zb, hi = za < zb ? zb : za;
          if (lo > zlo) { zlo = lo; face = k; nz = fabs(az); }
          if (hi < zup) zup = hi;
        }
        if (!inside || !(zlo <= zup) || zlo < -1e299) continue;
        const double sg = ax[face][2] > 0.0 ? 1.0 : -1.0;
        const double pos[3] = {x, y, zlo}, nn[3] = {sg * ax[face][0], sg * ax[face][1], sg * ax[face][2]};
        box_candidate(&A, (h - zlo) * nz, pos, nn);
      }
    /* (B) box edges x terrain edges */
    for (int k = 0; k < 3; ++k)
      for (int sb = 0; sb < 2; ++sb)
        for (int sc = 0; sc < 2; ++sc) {
          const int kb = (k + 1) % 3, kc = (k + 2) % 3;
          double p0[3], dir[3];
          for (int a = 0; a < 3; ++a) { p0[a] = ctr[a] - e[k][a] + (sb ? 1.0 : -1.0) * e[kb][a] + (sc ? 1.0 : -1.0) * e[kc][a]; dir[a] = 2.0 * e[k][a]; }
          const double gx0 = (p0[0] - x0) / dx, gy0 = (p0[1] - y0) / dy, dgx = dir[0] / dx, dgy = dir[1] / dy;
          for (int fam = 0; fam < 3; ++fam) {
            const double g0 = fam == 0 ? gx0 : (fam == 1 ? gy0 : gx0 - gy0), dg = fam == 0 ? dgx : (fam == 1 ? dgy : dgx - dgy);
            if (dg == 0.0) continue;
            const double g1 = g0 + dg;
            int lo = (int)ceil(g0 < g1 ? g0 : g1), hi = (int)floor(g0 < g1 ? g1 : g0);
            const int lim_lo = fam == 2 ? -(ys - 2) : 0, lim_hi = fam == 0 ? xs - 1 : (fam == 1 ? ys - 1 : xs - 2);
            if (lo < lim_lo) lo = lim_lo;
            if (hi > lim_hi) hi = lim_hi;
            if (hi - lo + 1 > ORC_BOX_SPAN) { hi = lo + ORC_BOX_SPAN - 1; *overflow = 1; }
            for (int i = lo; i <= hi; ++i) {
              double t = ((double)i - g0) / dg;
              t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
              const double pt[3] = {p0[0] + t * dir[0], p0[1] + t * dir[1], p0[2] + t * dir[2]};
              const double gx = (pt[0] - x0) / dx, gy = (pt[1] - y0) / dy;
              double hA, hB, f, T[3];
              if (fam == 0) {          /* grid line x = x0 + i dx: the terrain edge runs along y */
                if (gy < 0.0 || gy > ys - 1) continue;
                int j = (int)floor(gy); if (j > ys - 2) j = ys - 2;
                f = gy - j; hA = p->hm_heights[j * xs + i]; hB = p->hm_heights[(j + 1) * xs + i];
                T[0] = 0.0; T[1] = dy; T[2] = hB - hA;
              } else if (fam == 1) {   /* grid line y = y0 + i dy: along x */
                if (gx < 0.0 || gx > xs - 1) continue;
                int j = (int)floor(gx); if (j > xs - 2) j = xs - 2;
                f = gx - j; hA = p->hm_heights[i * xs + j]; hB = p->hm_heights[i * xs + j + 1];
                T[0] = dx; T[1] = 0.0; T[2] = hB - hA;
              } else {                 /* diagonal gx - gy = i: from vertex (jx, jx - i) to (jx + 1, jx - i + 1) */
                int jx = (int)floor(gx);
                if (jx > xs - 2) jx = xs - 2;
                if (jx < 0) jx = 0;
                const int jy = jx - i;
                f = gx - jx;
                if (jy < 0 || jy > ys - 2 || f < 0.0 || f > 1.0) continue;
                hA = p->hm_heights[jy * xs + jx]; hB = p->hm_heights[(jy + 1) * xs + jx + 1];
                T[0] = dx; T[1] = dy; T[2] = hB - hA;
              }
              const double h = hA + f * (hB - hA);
              double nn[3];
              cross3(dir, T, nn);
              const double n2 = dot3(nn, nn);
              if (!(n2 > 1e-12 * dot3(dir, dir) * dot3(T, T))) continue;    /* parallel edges: no crossing */
              const double inv = (nn[2] < 0.0 ? -1.0 : 1.0) / sqrt(n2);
              for (int a = 0; a < 3; ++a) nn[a] *= inv;
              box_candidate(&A, (h - pt[2]) * nn[2], pt, nn);
            }
          }
        }
  }
  if (!(A.sum[0] > 0.0 && A.dmax > 0.0 && A.dmax > dep_corners + ORC_CAPSULE_MARGIN)) return 0;
  const double nl = sqrt(A.sum[4] * A.sum[4] + A.sum[5] * A.sum[5] + A.sum[6] * A.sum[6]);
  for (int a = 0; a < 3; ++a) { c_out[a] = A.sum[1 + a] / A.sum[0]; n_out[a] = nl > 1e-9 ? A.sum[4 + a] / nl : (a == 2 ? 1.0 : 0.0); }
  *depth = A.dmax;
  return 1;
}

/* --------------------------------------------------------------------------- actuation */
/* PD controller, integrated implicitly ("stable PD", Tan, Liu, Turk 2011; RaiSim's controller is of this kind [RECALL]):
 * the torque the joint feels over the step is  kp (q* - q+) + kd (u* - u+)  with  q+ = q + dt u+.  Written for
 * the velocity update this is the explicit-looking torque
 *      tau = kp (q* - q - dt u) + kd (u* - u)
 * plus an extra joint-space inertia  B = dt (kd + dt kp)  on the diagonal of the mass matrix (returned in bdiag;
 * the contact problem sees the same effective mass).  Unconditionally stable in kp, kd; an effort-clipped joint is a
 * constant torque source (B = 0).  Joint damping stays explicit. */
static void actuation_impl(const rsb_model_blob* m, const orc_params* p, const double* q, const double* u,
                           const double* kp, const double* kd, const double* p_target,
                           const double* d_target, const double* tau_ff, double* tau, double* bdiag) {
  for (int d = 0; d < 6; ++d) { tau[d] = tau_ff ? tau_ff[d] : 0.0; if (bdiag) bdiag[d] = 0.0; }
  for (int i = 1; i < m->nb; ++i) {
    int d = dof_of(i), qi = qidx_of(i);
    double t = tau_ff ? tau_ff[d] : 0.0, B = 0.0;
    if (p->control_mode == RSB_PD_PLUS_FEEDFORWARD_TORQUE && kp && kd) {
      double pt = p_target ? p_target[qi] : 0.0, dt_ = d_target ? d_target[d] : 0.0;
      t += kp[d] * (pt - q[qi] - p->dt * u[d]) + kd[d] * (dt_ - u[d]);
      B = p->dt * (kd[d] + p->dt * kp[d]);
    }
    if (m->effort[i] > 0 && fabs(t) > m->effort[i]) { t = t > 0 ? m->effort[i] : -m->effort[i]; B = 0.0; }
    tau[d] = t - m->damping[i] * u[d];
    if (bdiag) bdiag[d] = B;
  }
}
/* the torque part alone (tests) */
void orc_actuation(const rsb_model_blob* m, const orc_params* p, const double* q, const double* u,
                   const double* kp, const double* kd, const double* p_target,
                   const double* d_target, const double* tau_ff, double* tau) {
  actuation_impl(m, p, q, u, kp, kd, p_target, d_target, tau_ff, tau, NULL);
}

/* ------------------------------------------------------------------- per-contact solver */
/*
 * Slip case of one contact (Hwangbo et al. 2018 §III-B).  With the stick impulse lam_s = -G^-1 v outside the
 * friction cone, the impulse is the point of  C = {lam in cone} ∩ {v_n^+ = 0}  that minimises the contact-space
 * kinetic energy  E = 1/2 (lam - lam_s)^T G (lam - lam_s)  (= 1/2 v^+T G^-1 v^+).  E is strictly convex and C is
 * convex, so the minimiser is unique and lies on the boundary curve
 *      lam(d) = ln(d) (mu d, 1),   ln(d) = -v_n / (G_nn + mu G_nt.d),   |d| = 1,
 * which is an ellipse when mu |G_nt| < G_nn and unbounded otherwise (E -> inf towards the asymptote, so the
 * minimiser stays finite even in Painleve-type jamming configurations).  Since frictionless inelastic impact is
 * a member of C, the result never increases the kinetic energy.
 *
 * slip_E  : E along direction d (+inf where G_nn + mu G_nt.d <= 0: no point of the curve in that direction)
 * slip_dE : a positive multiple of dE/dtheta along the curve (theta = angle of d), used only for its sign:
 *           dE/dtheta = v^+ . dlam/dtheta = mu ln [ v_t^+.dperp - (den'/den) v_t^+.d ],  den' = mu G_nt.dperp
 */
#define ORC_DEN_MIN 1e-6
#define ORC_DEN_FREEZE 0.1
static const double kCos16[16] = {1.0, 0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0.0,
                                  -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                                  -0.92387953251128674, -0.70710678118654752, -0.38268343236508977, 0.0,
                                  0.38268343236508977, 0.70710678118654752, 0.92387953251128674};
static const double kSin16[16] = {0.0, 0.38268343236508977, 0.70710678118654752, 0.92387953251128674, 1.0,
                                  0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0.0,
                                  -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                                  -0.92387953251128674, -0.70710678118654752, -0.38268343236508977};

/* Both are evaluated through 9 coefficients that are constant during one slip solve:
 *   den(d) = a0 + a1 x + a2 y,   N(d) := den * v_t^+ = [n00 + n01 x + n02 y ; n10 + n11 x + n12 y]
 * so a candidate costs a handful of FMAs and slip_dE needs no division (its value is dE/dtheta times den^2 / (mu ln),
 * a positive factor).  The device computes the coefficients once per solve on the contact's own lane. */
typedef struct slip_coef { double a0, a1, a2, n00, n01, n02, n10, n11, n12, vn, ls0, ls1, mu, bx, by; int coul; } slip_coef;

static void slip_prepare(const double* G, const double* v, const double* ls, double mu, slip_coef* k) {
  k->a0 = G[8]; k->a1 = mu * G[6]; k->a2 = mu * G[7];
  k->n00 = k->a0 * v[0] - v[2] * G[2]; k->n01 = k->a1 * v[0] - v[2] * mu * G[0]; k->n02 = k->a2 * v[0] - v[2] * mu * G[1];
  k->n10 = k->a0 * v[1] - v[2] * G[5]; k->n11 = k->a1 * v[1] - v[2] * mu * G[3]; k->n12 = k->a2 * v[1] - v[2] * mu * G[4];
  k->vn = v[2]; k->ls0 = ls[0]; k->ls1 = ls[1]; k->mu = mu; k->coul = 0;
}
static double slip_E(const slip_coef* k, double x, double y) {
  double den = k->a0 + k->a1 * x + k->a2 * y;
  if (!(den > ORC_DEN_MIN * k->a0)) return 1e300;
  double inv = 1.0 / den, ln = -k->vn * inv;
  double vt0 = (k->n00 + k->n01 * x + k->n02 * y) * inv, vt1 = (k->n10 + k->n11 * x + k->n12 * y) * inv;
  return 0.5 * (vt0 * (k->mu * ln * x - k->ls0) + vt1 * (k->mu * ln * y - k->ls1));
}
/* k->coul != 0: the CLASSICAL COULOMB rule (orc_params::slip_rule) - the root of  P(theta) = N x d = den v_t+ x d  (slip velocity parallel to the
 * impulse direction) that P crosses upwards, where v_t+ . d < 0: the same formulas with (den, a0, mdp) replaced by (1, 1, 0) */
static double slip_dE(const slip_coef* k, double x, double y) {
  double den = k->a0 + k->a1 * x + k->a2 * y;
  double mdp = k->a2 * x - k->a1 * y;                 /* mu * G_nt . dperp, dperp = (-y, x) */
  /* no curve point in this direction: the infeasible arc is contiguous, < 180 deg and does not contain the
   * round-0 best direction b, so the minimiser lies on b's side of the candidate */
  if (!(den > ORC_DEN_MIN * k->a0)) return (k->bx * y - k->by * x > 0.0) ? 1.0 : -1.0;
  double N0 = k->n00 + k->n01 * x + k->n02 * y, N1 = k->n10 + k->n11 * x + k->n12 * y;
  if (k->coul) return N1 * x - N0 * y;
  return den * (N1 * x - N0 * y) - mdp * (N0 * x + N1 * y);
}

/* Local refinement of a previous slip direction (x0, y0): one Newton step on h(theta) = slip_dE along the curve,
 *   h' = den (N1' x - N0' y) - a0 (N0 x + N1 y) - mdp (N0' x + N1' y),   N' = dN/dtheta = n.2 x - n.1 y,
 * accepted only when it is a safe descent step: direction well inside the feasible arc before and after, h' > 0
 * (a minimum, not a maximum, nearby), |dtheta| <= 0.25 rad, and for steps above 0.02 rad no energy increase.
 * Returns 0 when rejected (the caller then runs the global search). */
#define ORC_DEN_NEWTON 1e-3
#define ORC_POLISH_STEPS 2
#ifdef ORC_STATS
long orc_stats[8];   /* [0] newton accepted, [1..5] rejected at den0 / hp / |d| / den1 / E, [6] global searches */
#define ORC_STAT(i) (++orc_stats[i])
#else
#define ORC_STAT(i) ((void)0)
#endif
/* Newton step of h(theta) = slip_dE at the unit direction (x0, y0): returns dtheta = -h / h' and h' through *hp */
static double slip_newton_step(const slip_coef* k, double x0, double y0, double* hp) {
  double den = k->a0 + k->a1 * x0 + k->a2 * y0;
  double mdp = k->a2 * x0 - k->a1 * y0;
  double N0 = k->n00 + k->n01 * x0 + k->n02 * y0, N1 = k->n10 + k->n11 * x0 + k->n12 * y0;
  double dN0 = k->n02 * x0 - k->n01 * y0, dN1 = k->n12 * x0 - k->n11 * y0;
  double P = N1 * x0 - N0 * y0, Q = N0 * x0 + N1 * y0;
  if (k->coul) { *hp = (dN1 * x0 - dN0 * y0) - Q; return -P / *hp; }      /* Coulomb: h = P */
  double h = den * P - mdp * Q;
  *hp = den * (dN1 * x0 - dN0 * y0) - k->a0 * Q - mdp * (dN0 * x0 + dN1 * y0);
  return -h / *hp;
}
/* |P| and Q = N . d at a unit direction (the Coulomb rule's residual and the sign of the slip along the impulse) */
static void slip_PQ(const slip_coef* k, double x, double y, double* P, double* Q) {
  double N0 = k->n00 + k->n01 * x + k->n02 * y, N1 = k->n10 + k->n11 * x + k->n12 * y;
  *P = N1 * x - N0 * y; *Q = N0 * x + N1 * y;
}
/* (x0, y0) rotated by the small angle d (|d| <= 0.25: degree-5 Taylor polynomials), renormalised */
static void slip_rotate(double x0, double y0, double d, double* x1, double* y1) {
  double d2 = d * d;
  double c = 1.0 - d2 * (0.5 - d2 * (1.0 / 24.0)), s = d * (1.0 - d2 * ((1.0 / 6.0) - d2 * (1.0 / 120.0)));
  double x = x0 * c - y0 * s, y = x0 * s + y0 * c, inv = 1.0 / sqrt(x * x + y * y);
  *x1 = x * inv; *y1 = y * inv;
}
static int slip_newton(const slip_coef* k, double x0, double y0, double* x1, double* y1, double* step) {
  double den = k->a0 + k->a1 * x0 + k->a2 * y0, hp;
  if (!(den > ORC_DEN_NEWTON * k->a0)) { ORC_STAT(1); return 0; }
  double d = slip_newton_step(k, x0, y0, &hp);
  if (!(hp > 0.0)) { ORC_STAT(2); return 0; }
  if (!(fabs(d) <= 0.25)) { ORC_STAT(3); return 0; }
  double x, y;
  slip_rotate(x0, y0, d, &x, &y);
  if (!(k->a0 + k->a1 * x + k->a2 * y > ORC_DEN_NEWTON * k->a0)) { ORC_STAT(4); return 0; }
  if (k->coul) {      /* Coulomb: the slip must oppose the impulse at the new direction, and a large step must reduce the residual */
    double P0, Q0, P1, Q1;
    slip_PQ(k, x0, y0, &P0, &Q0); slip_PQ(k, x, y, &P1, &Q1);
    if (!(Q1 < 0.0) || (fabs(d) > 0.02 && !(fabs(P1) <= fabs(P0)))) { ORC_STAT(5); return 0; }
  } else
  if (fabs(d) > 0.02 && !(slip_E(k, x, y) <= slip_E(k, x0, y0))) { ORC_STAT(5); return 0; }
  ORC_STAT(0);
  *x1 = x; *y1 = y; *step = d;
  return 1;
}

/*
 * One contact of the per-contact iteration: given the contact-space velocity v the contact would have with its
 * own impulse removed, and its own 3x3 Delassus block G (contact frame [t1 t2 n]), return the impulse:
 *   open : v_n > 0                         -> 0
 *   stick: lam_s = -G^-1 v inside the cone -> lam_s
 *   slip : minimum-energy point of the curve above, located by
 *            round 0 : E at 16 directions 22.5 deg apart; the best one +-1 neighbour brackets the minimiser;
 *            rounds 1..section_rounds : 16-section on the sign of dE/dtheta: 15 candidates on the chord between
 *                      the bracket ends (normalised), the first candidate with dE >= 0 closes the bracket from
 *                      above; the new ends are the (un-normalised) chord points -> 45deg / 16^2 = 3e-3 rad;
 *            polish  : two Newton steps on dE/dtheta from the bracket midpoint, clamped to the bracket
 *                      (quadratic convergence: 1.5e-3 -> ~1e-6 -> ~1e-12 rad).
 *          16-section instead of bisection because the device evaluates the 15 (16) candidates of a round on the
 *          lanes of the env group at once; the oracle walks the same candidates sequentially.
 */
/* sdir (in/out, 3 doubles: dx, dy, valid): the friction direction of this contact's last slip solve.  With
 * use_frozen != 0 and a valid direction the slip case keeps that direction and only re-solves the magnitude
 * ("lagged friction direction", used by the caller after `freeze_after` sweeps, and from the sweep after a Newton
 * refinement moved the direction by less than settle_tol rad: sdir[2] = 2 marks such a settled direction).  With refine != 0 and a valid
 * direction the global search is replaced by slip_newton() whenever that step is accepted. */
/* coulomb != 0 (orc_params::slip_rule = ORC_SLIP_COULOMB): the slip case looks for the CLASSICAL COULOMB point of the same curve instead of the
 * least-energy one - the direction where the post-impulse slip velocity is anti-parallel to the friction impulse:
 *   round 0 : P = den v_t+ x d at the 16 grid directions; an interval [k, k+1] of feasible directions with P_k < 0 <= P_k+1 holds such a root
 *             (upward crossing: there v_t+ . d < 0; the downward crossings are the roots where friction would PUSH the contact along its slip);
 *             of several the one whose lower end has the least energy; NONE (two roots inside one 22.5 deg interval, or no feasible crossing):
 *             the energy rule's search for this solve;
 *   rounds 1.. and the polish: the same 16-section and Newton steps on P instead of dE/dtheta;
 *   refinement of an earlier direction: one Newton step on P, accepted when the slip opposes the impulse there (and a step above 0.02 rad
 *             reduces |P|); an inherited direction needs no basin check (it was a root of the previous step's problem). */
static void solve_one_contact(const double* G, const double* Ginv, const double* v, double mu,
                              int section_rounds, int use_frozen, int refine, double settle_tol, double* sdir, double* lam, int coulomb) {
  if (v[2] > 0.0) { lam[0] = lam[1] = lam[2] = 0.0; return; }
  double ls[3];
  for (int r = 0; r < 3; ++r) ls[r] = -(Ginv[3 * r] * v[0] + Ginv[3 * r + 1] * v[1] + Ginv[3 * r + 2] * v[2]);
  double lt2 = ls[0] * ls[0] + ls[1] * ls[1];
  if (ls[2] >= 0.0 && lt2 <= mu * mu * ls[2] * ls[2]) { lam[0] = ls[0]; lam[1] = ls[1]; lam[2] = ls[2]; return; }
  slip_coef k;
  slip_prepare(G, v, ls, mu, &k);
  k.coul = coulomb;
  /* sdir[2]: 0 no direction, 1 direction of an earlier slip solve of THIS integrate(), 2 the same and settled,
   * 3 inherited from the previous integrate() through the warm state and not yet used in this one */
  const int inherited = sdir[2] == 3.0;
  if ((use_frozen || sdir[2] == 2.0) && sdir[2] != 0.0 && !inherited) {
    /* only well-conditioned directions are kept: near the curve's asymptote (den -> 0) a stale direction would
     * amplify any change of v_n without bound */
    double den = k.a0 + k.a1 * sdir[0] + k.a2 * sdir[1];
    if (den >= ORC_DEN_FREEZE * k.a0) {
      double ln = -v[2] / den;
      lam[0] = mu * ln * sdir[0]; lam[1] = mu * ln * sdir[1]; lam[2] = ln;
      return;
    }
  }
  if (refine && sdir[2] != 0.0) {
    double x, y, d;
    int ok = slip_newton(&k, sdir[0], sdir[1], &x, &y, &d);
    if (ok && inherited && !coulomb) {
      /* basin check: E restricted to the curve can have two local minima, and a direction carried over from the previous
       * time step may sit in the one the global search would not choose (measured: 1 solve in 24 000 of the config-2
       * population, 0.3 m/s off).  The refined direction is accepted only if it is at least as good as every direction
       * of the search's coarse scan; otherwise the global search runs. */
      double ebest = slip_E(&k, kCos16[0], kSin16[0]);
      for (int i = 1; i < 16; ++i) { double e = slip_E(&k, kCos16[i], kSin16[i]); if (e < ebest) ebest = e; }
      if (!(slip_E(&k, x, y) <= ebest)) ok = 0;
    }
    if (ok) {
      double ln = -v[2] / (k.a0 + k.a1 * x + k.a2 * y);
      lam[0] = mu * ln * x; lam[1] = mu * ln * y; lam[2] = ln;
      sdir[0] = x; sdir[1] = y;
      sdir[2] = (settle_tol > 0.0 && fabs(d) <= settle_tol) ? 2.0 : 1.0;   /* settled: the direction stopped moving, later sweeps keep it */
      return;
    }
  }
  ORC_STAT(6);
  int kbest = 0;
  double ebest = slip_E(&k, kCos16[0], kSin16[0]);
  for (int i = 1; i < 16; ++i) {
    double e = slip_E(&k, kCos16[i], kSin16[i]);
    if (e < ebest) { ebest = e; kbest = i; }
  }
  double lox = kCos16[(kbest + 15) & 15], loy = kSin16[(kbest + 15) & 15];
  double hix = kCos16[(kbest + 1) & 15], hiy = kSin16[(kbest + 1) & 15];
  if (coulomb) {
    int kc = -1;
    double ec = 1e300;
    for (int i = 0; i < 16; ++i) {
      const int j = (i + 1) & 15;
      const double d0 = k.a0 + k.a1 * kCos16[i] + k.a2 * kSin16[i], d1 = k.a0 + k.a1 * kCos16[j] + k.a2 * kSin16[j];
      if (!(d0 > ORC_DEN_MIN * k.a0) || !(d1 > ORC_DEN_MIN * k.a0)) continue;
      double P0, Q0, P1, Q1;
      slip_PQ(&k, kCos16[i], kSin16[i], &P0, &Q0); slip_PQ(&k, kCos16[j], kSin16[j], &P1, &Q1);
      if (!(P0 < 0.0 && P1 >= 0.0)) continue;
      if (!(Q0 < 0.0 && Q1 < 0.0)) continue;      /* the slip opposes the impulse on the whole interval (with coupling an upward crossing alone does not say so) */
      const double e = slip_E(&k, kCos16[i], kSin16[i]);
      if (e < ec) { ec = e; kc = i; }
    }
    if (kc >= 0) { kbest = kc; lox = kCos16[kc]; loy = kSin16[kc]; hix = kCos16[(kc + 1) & 15]; hiy = kSin16[(kc + 1) & 15]; }
    else k.coul = 0;      /* no bracketed Coulomb root: the energy rule for this solve */
  }
  if (coulomb && k.coul) { k.bx = lox + hix; k.by = loy + hiy; }      /* (a direction inside the bracket: the side rule of slip_dE) */
  else { k.bx = kCos16[kbest]; k.by = kSin16[kbest]; }
  for (int r = 0; r < section_rounds; ++r) {
    double ex = hix - lox, ey = hiy - loy;
    int kstar = 15;
    for (int i = 0; i < 15; ++i) {
      double t = (i + 1) * (1.0 / 16.0);
      double x = lox + t * ex, y = loy + t * ey, inv = 1.0 / sqrt(x * x + y * y);
      if (slip_dE(&k, x * inv, y * inv) >= 0.0) { kstar = i; break; }
    }
    double tl = kstar * (1.0 / 16.0), th = tl + (1.0 / 16.0);
    double nlx = lox + tl * ex, nly = loy + tl * ey, nhx = lox + th * ex, nhy = loy + th * ey;
    if (kstar < 15) { hix = nhx; hiy = nhy; }
    if (kstar > 0) { lox = nlx; loy = nly; }
  }
  /* polish: two Newton steps from the bracket midpoint, each clamped to the bracket's half width */
  double mx = lox + hix, my = loy + hiy, ex = hix - lox, ey = hiy - loy;
  double im = 1.0 / sqrt(mx * mx + my * my), w = sqrt(ex * ex + ey * ey) * im;
  double x = mx * im, y = my * im;
  for (int r = 0; r < ORC_POLISH_STEPS; ++r) {
    double hp, d = slip_newton_step(&k, x, y, &hp);
    if (!(hp > 0.0)) d = 0.0;
    if (d > w) d = w;
    if (d < -w) d = -w;
    slip_rotate(x, y, d, &x, &y);
  }
  double den = k.a0 + k.a1 * x + k.a2 * y;
  if (!(den > ORC_DEN_MIN * k.a0)) den = ORC_DEN_MIN * k.a0;
  double ln = -v[2] / den;
  lam[0] = mu * ln * x; lam[1] = mu * ln * y; lam[2] = ln;
  sdir[0] = x; sdir[1] = y; sdir[2] = 1.0;
}

/* one contact solved in isolation (unit tests of the open/stick/slip rule against a dense minimisation) */
void orc_solve_contact(const double* G, const double* v, double mu, int section_rounds, double* lam) {
  double Ginv[9], sdir[3] = {0.0, 0.0, 0.0};
  inv3(G, Ginv);
  solve_one_contact(G, Ginv, v, mu, section_rounds, 0, 0, 0.0, sdir, lam, 0);
}
/* ... with the slip rule chosen (ORC_SLIP_ENERGY / ORC_SLIP_COULOMB); *used_rule (may be NULL) receives the rule the slip case ended up with
 * (a Coulomb solve without a bracketed root falls back to the energy rule) */
void orc_solve_contact_rule(const double* G, const double* v, double mu, int section_rounds, int rule, double* lam) {
  double Ginv[9], sdir[3] = {0.0, 0.0, 0.0};
  inv3(G, Ginv);
  solve_one_contact(G, Ginv, v, mu, section_rounds, 0, 0, 0.0, sdir, lam, rule);
}

static void contact_frame(const double* n, double* Rc /* columns t1 t2 n, row-major */) {
  /* t1 = the normalised projection of a world axis on the tangent plane: world x, or world y when the normal is (nearly)
   * along x - a self-collision between mirror-symmetric limbs has n = (+-1, 0, 0) exactly, and the projection of x vanishes */
  double t1[3], t2[3];
  const int ry = fabs(n[0]) > 0.9;
  double dn = ry ? n[1] : n[0];
  t1[0] = (ry ? 0.0 : 1.0) - dn * n[0]; t1[1] = (ry ? 1.0 : 0.0) - dn * n[1]; t1[2] = -dn * n[2];
  double il = 1.0 / sqrt(dot3(t1, t1));
  for (int c = 0; c < 3; ++c) t1[c] *= il;
  cross3(n, t1, t2);
  for (int c = 0; c < 3; ++c) { Rc[3 * c] = t1[c]; Rc[3 * c + 1] = t2[c]; Rc[3 * c + 2] = n[c]; }
}

/* ---------------------------------------------------------------------------------- self-collision */
/* Candidate pairs (RaiSim collides the links of one articulated system with each other, parent-child pairs excepted [RECALL]):
 * primitives i < j on two different bodies that are not parent and child, not both points, not rim primitives (their point
 * is defined against the terrain only), and whose bodies are not in the caller's ignore set (ignoreCollisionBetween). */
static int self_pair_ok(const rsb_model_blob* m, const uint8_t* ignore, int i, int j) {
  const int bi = m->col_body[i], bj = m->col_body[j];
  if (bi == bj || m->parent[bi] == bj || m->parent[bj] == bi) return 0;
  if (m->col_rim[i] > 0.0 || m->col_rim[j] > 0.0) return 0;
  if (!(m->col_radius[i] + m->col_radius[j] > 0.0)) return 0;
  if (ignore && (ignore[bi * m->nb + bj] || ignore[bj * m->nb + bi])) return 0;
  return 1;
}

int orc_self_pairs(const rsb_model_blob* m, const uint8_t* ignore, int32_t* pairs, int cap) {
  int n = 0;
  for (int i = 0; i < m->ncol; ++i)
    for (int j = i + 1; j < m->ncol; ++j)
      if (self_pair_ok(m, ignore, i, j)) {
        if (pairs && n < cap) { pairs[2 * n] = i; pairs[2 * n + 1] = j; }
        ++n;
      }
  return n;
}

/* ---------------------------------------------------------------------------------- step */
static void step_impl(const rsb_model_blob* m, const orc_params* p, double* q, double* u, const double* kp,
                      const double* kd, const double* p_target, const double* d_target,
                      const double* tau_ff, orc_contact* contacts, int32_t* n_contacts, int32_t* iters,
                      int32_t* flags, double* lam_warm, double* dbgG, double* dbgc, double* dbglam) {
  int nv = m->nv, nq = m->nq, kmax = p->kmax > MAXK ? MAXK : p->kmax;
  /* per-thread scratch (no malloc in the stepping loop: this function is also the timed CPU baseline) */
  static _Thread_local kin_t* tl_k = NULL;
  static _Thread_local double* tl_M = NULL;
  static _Thread_local double* tl_JX = NULL;
  if (!tl_k) {
    tl_k = (kin_t*)malloc(sizeof(kin_t));
    tl_M = (double*)malloc(sizeof(double) * MAXV * MAXV);
    tl_JX = (double*)malloc(sizeof(double) * (2 * MAXK * 3 * MAXV + 3 * MAXV));
  }
  kin_t* k = tl_k;
  double* M = tl_M;
  double h[MAXV], tau[MAXV], ufree[MAXV];
  int pd[MAXV], fl = 0;

  if (m->fixed_base) for (int d = 0; d < 6; ++d) u[d] = 0.0;   /* a fixed base has no velocity, whatever the caller's row says */
  /* integrate1: kinematics, collision detection, M, h */
  kinematics(m, q, u, k);
  crba(m, k, M);
  rnea(m, k, u, NULL, p->gravity, h);
  {
    double bdiag[MAXV];
    actuation_impl(m, p, q, u, kp, kd, p_target, d_target, tau_ff, tau, bdiag);
    for (int d = 6; d < nv; ++d) M[d * nv + d] += bdiag[d];   /* implicit PD: see actuation_impl */
    if (m->fixed_base) for (int d = 0; d < 6; ++d) M[d * nv + d] += 1e30;   /* fixed base = a base of (numerically) infinite inertia */
  }

  int nc = 0;
  double cx[MAXK][3], cn[MAXK][3], cdepth[MAXK], Rc[MAXK][9];
  int cbody[MAXK], ccol[MAXK];
  int cbody2[MAXK], ccol2[MAXK];   /* second body / primitive of a self-collision (-1: contact with the terrain) */
  double cmat[MAXK][3];            /* mu, restitution, threshold of a self-collision's material pair */
  double cen[RSB_MAX_COLLISIONS][3];   /* primitive centres relative to the base position */
  int csecond[MAXK];               /* second contact of a primitive with the terrain (a valley's other flank) */
  double sec_depth[RSB_MAX_COLLISIONS], sec_n[RSB_MAX_COLLISIONS][3], sec_c[RSB_MAX_COLLISIONS][3];
  double first_depth[RSB_MAX_COLLISIONS];   /* penetration of each primitive's first contact (0: none) */
  double cap_c[RSB_MAX_COLLISIONS][3];      /* primitive centres as the model gives them (a rim primitive's cen[] is the lowest point of its rim) */
  for (int i = 0; i < MAXK; ++i) { cbody2[i] = -1; ccol2[i] = -1; csecond[i] = 0; }
  for (int s = 0; s < m->ncol; ++s) {
    int b = m->col_body[s];
    double t[3], c[3], cw[3], n[3], depth;
    mat3_vec(k->R[b], m->col_pos[s], t);
    for (int a = 0; a < 3; ++a) { c[a] = k->r[b][a] + t[a]; cap_c[s][a] = c[a]; }
    if (m->col_rim[s] > 0.0) {
      /* rim primitive (end cap of a cylinder): the point of the circle of radius col_rim around c, normal to the cap's axis,
       * that is lowest along the world's vertical: c - rim * e / |e| with e = z - (z.a) a; a cap lying flat keeps its centre */
      double aw[3];
      mat3_vec(k->R[b], m->col_axis[s], aw);
      const double len2 = 1.0 - aw[2] * aw[2];
      if (len2 > 1e-12) {
        const double kk = m->col_rim[s] / sqrt(len2);
        c[0] += kk * aw[2] * aw[0]; c[1] += kk * aw[2] * aw[1]; c[2] -= kk * len2;
      }
    }
    for (int a = 0; a < 3; ++a) { cw[a] = k->pbase[a] + c[a]; cen[s][a] = c[a]; }
    double depth2 = 0.0, n2[3] = {0, 0, 1};
    const int hit = terrain_contact(p, cw, m->col_radius[s], &depth, n, &depth2, n2);
    sec_depth[s] = hit ? depth2 : 0.0; for (int a = 0; a < 3; ++a) { sec_n[s][a] = n2[a]; sec_c[s][a] = c[a]; }
    first_depth[s] = hit ? depth : 0.0;
    if (hit) {
      if (nc >= kmax) { fl |= 1; continue; }
      for (int a = 0; a < 3; ++a) { cx[nc][a] = c[a] - m->col_radius[s] * n[a]; cn[nc][a] = n[a]; }
      cdepth[nc] = depth; cbody[nc] = b; ccol[nc] = s;
      contact_frame(n, Rc[nc]);
      ++nc;
    }
  }
  /* the second flank of a valley (orc_params::hm_contacts): after all first contacts, in primitive order (the device emits them in a
   * second pass over the primitives); its own slot, the primitive's material, a cold start */
  for (int s = 0; s < m->ncol; ++s) {
    if (!(sec_depth[s] > 0.0)) continue;
    if (nc >= kmax) { fl |= 1; continue; }
    for (int a = 0; a < 3; ++a) { cx[nc][a] = sec_c[s][a] - m->col_radius[s] * sec_n[s][a]; cn[nc][a] = sec_n[s][a]; }
    cdepth[nc] = sec_depth[s]; cbody[nc] = m->col_body[s]; ccol[nc] = s; csecond[nc] = 1;
    contact_frame(sec_n[s], Rc[nc]);
    ++nc;
  }
  /* the cylinders of the capsules (orc_params::hm_capsule), after the second flanks, in primitive order; a slot of its own, the first
   * end sphere's material, a cold start (csecond = 2) */
  if (p->hm_capsule && p->terrain_type == 1) {
    for (int s = 0; s < m->ncol; ++s) {
      if (m->col_capsule[s] == 0) continue;
      if (m->col_capsule[s] == -1) {       /* a box: the deepest point of its faces */
        double cwb[8][3], cc[3], n[3], depth, dep_c = 0.0;
        for (int e = 0; e < 8; ++e) { for (int a = 0; a < 3; ++a) cwb[e][a] = k->pbase[a] + cap_c[s + e][a]; if (first_depth[s + e] > dep_c) dep_c = first_depth[s + e]; }
        int over = 0;
        const int got = box_face_contact(p, cwb, dep_c, cc, &depth, n, &over);
        if (over) fl |= 1;
        if (!got) continue;
        if (nc >= kmax) { fl |= 1; continue; }
        for (int a = 0; a < 3; ++a) { cx[nc][a] = cc[a] - k->pbase[a]; cn[nc][a] = n[a]; }
        cdepth[nc] = depth; cbody[nc] = m->col_body[s]; ccol[nc] = s; csecond[nc] = 2;
        contact_frame(n, Rc[nc]);
        ++nc;
        continue;
      }
      const int e = m->col_capsule[s] - 1;
      double aw[3], bw[3], cc[3], n[3], depth;
      for (int a = 0; a < 3; ++a) { aw[a] = k->pbase[a] + cap_c[s][a]; bw[a] = k->pbase[a] + cap_c[e][a]; }
      const double dep_ends = first_depth[s] > first_depth[e] ? first_depth[s] : first_depth[e];
      const int cyl = m->col_rim[s] > 0.0;            /* a cylinder: radius = its rims', samples stay r / L away from the flat caps */
      const double rr = cyl ? m->col_rim[s] : m->col_radius[s];
      const double len = sqrt((bw[0] - aw[0]) * (bw[0] - aw[0]) + (bw[1] - aw[1]) * (bw[1] - aw[1]) + (bw[2] - aw[2]) * (bw[2] - aw[2]));
      if (!capsule_contact(p, aw, bw, rr, cyl ? rr / len : 0.02, dep_ends, cc, &depth, n)) continue;
      if (nc >= kmax) { fl |= 1; continue; }
      for (int a = 0; a < 3; ++a) { cx[nc][a] = cc[a] - k->pbase[a] - rr * n[a]; cn[nc][a] = n[a]; }
      cdepth[nc] = depth; cbody[nc] = m->col_body[s]; ccol[nc] = s; csecond[nc] = 2;
      contact_frame(n, Rc[nc]);
      ++nc;
    }
  }

  /* Self-collision (sphere x sphere): two primitives of the candidate set closer than r_i + r_j touch in the middle of the
   * overlap, normal from j to i.  ONE contact in the solver (J = J_i - J_j), TWO entries in the contact list (one per body,
   * opposite normals and impulses, as RaiSim lists it) - and two of the kmax slots, as on the device. */
  int nslots = nc, nself = 0;
  if (p->self_collision) {
    int pair = -1;
    for (int i = 0; i < m->ncol; ++i)
      for (int j = i + 1; j < m->ncol; ++j) {
        if (!self_pair_ok(m, p->self_ignore, i, j)) continue;
        ++pair;
        double d[3] = {cen[i][0] - cen[j][0], cen[i][1] - cen[j][1], cen[i][2] - cen[j][2]};
        const double rs = m->col_radius[i] + m->col_radius[j], d2 = dot3(d, d);
        if (!(d2 < rs * rs) || d2 < 1e-12) continue;
        if (nslots + 2 > kmax) { fl |= 1; continue; }
        const double dist = sqrt(d2), depth = rs - dist;
        double n[3] = {d[0] / dist, d[1] / dist, d[2] / dist};
        for (int a = 0; a < 3; ++a) { cx[nc][a] = cen[i][a] - (m->col_radius[i] - 0.5 * depth) * n[a]; cn[nc][a] = n[a]; }
        cdepth[nc] = depth; cbody[nc] = m->col_body[i]; ccol[nc] = i; cbody2[nc] = m->col_body[j]; ccol2[nc] = j;
        cmat[nc][0] = p->self_mu ? p->self_mu[pair] : p->mu;
        cmat[nc][1] = p->self_restitution ? p->self_restitution[pair] : p->restitution;
        cmat[nc][2] = p->self_res_threshold ? p->self_res_threshold[pair] : p->res_threshold;
        contact_frame(n, Rc[nc]);
        ++nc; ++nself; nslots += 2;
      }
  }

  /* joint limits (RaiSim enforces them in the same solver [RECALL]): a joint beyond its range adds one unilateral row
   * s * qdot >= 0 (s = -1 above the upper limit, +1 below the lower one), carried through the solver as a contact whose
   * two tangential rows are empty (unit dummy diagonal); it is not reported by getContacts */
  int nreal = nc;   /* (not const: the contact-set reduction experiment runs the solver on a compacted set and restores it) */
  double lim_sign[MAXK];
  for (int i = 0; i < nc; ++i) lim_sign[i] = 0.0;
  for (int i = 1; i < m->nb; ++i) {
    const double lo = m->q_lower[i], hi = m->q_upper[i], qi = q[qidx_of(i)];
    if (!(lo < hi)) continue;
    double sgn = 0.0, viol = 0.0;
    if (qi > hi) { sgn = -1.0; viol = qi - hi; } else if (qi < lo) { sgn = 1.0; viol = lo - qi; }
    if (sgn != 0.0) {
      if (nslots >= kmax) { fl |= 1; continue; }
      ++nslots;
      cbody[nc] = i; ccol[nc] = m->ncol + i; cdepth[nc] = viol; lim_sign[nc] = sgn;
      for (int a = 0; a < 3; ++a) { cx[nc][a] = 0.0; cn[nc][a] = 0.0; }
      ++nc;
    }
  }

  /* integrate2: u_free = u + dt M^-1 (tau - h) */
  dof_parents(m, pd);
  ltdl(M, nv, pd);
  for (int i = 0; i < nv; ++i) ufree[i] = p->dt * (tau[i] - h[i]);
  ltdl_solve(M, nv, pd, ufree);
  for (int i = 0; i < nv; ++i) ufree[i] += u[i];

  double lam[MAXK][3];
  double sdir_out[MAXK][3];   /* friction directions at the end of the solve (warm state of the next integrate()) */
  for (int i = 0; i < MAXK; ++i) sdir_out[i][0] = sdir_out[i][1] = sdir_out[i][2] = 0.0;
  int it_used = 0;
  double (*X)[3][MAXV] = NULL;
  if (nc > 0) {
    double (*Jc)[3][MAXV] = (double (*)[3][MAXV])tl_JX;
    X = (double (*)[3][MAXV])(tl_JX + MAXK * 3 * MAXV);
    double* Jw = tl_JX + 2 * MAXK * 3 * MAXV;
    for (int i = 0; i < nc; ++i) {
      if (lim_sign[i] != 0.0) {   /* joint-limit row: J = [0; 0; s e_k] */
        for (int r = 0; r < 3; ++r) for (int d = 0; d < nv; ++d) { Jc[i][r][d] = 0.0; X[i][r][d] = 0.0; }
        Jc[i][2][dof_of(cbody[i])] = lim_sign[i]; X[i][2][dof_of(cbody[i])] = lim_sign[i];
        ltdl_solve(M, nv, pd, X[i][2]);
        continue;
      }
      point_jacobian(m, k, cbody[i], cx[i], Jw);
      for (int r = 0; r < 3; ++r)
        for (int d = 0; d < nv; ++d) {
          double s = 0;
          for (int c = 0; c < 3; ++c) s += Rc[i][3 * c + r] * Jw[c * nv + d];
          Jc[i][r][d] = s;
        }
      if (cbody2[i] >= 0) {   /* self-collision: relative velocity of the two bodies' points */
        point_jacobian(m, k, cbody2[i], cx[i], Jw);
        for (int r = 0; r < 3; ++r)
          for (int d = 0; d < nv; ++d) {
            double s = 0;
            for (int c = 0; c < 3; ++c) s += Rc[i][3 * c + r] * Jw[c * nv + d];
            Jc[i][r][d] -= s;
          }
      }
      for (int r = 0; r < 3; ++r) for (int d = 0; d < nv; ++d) X[i][r][d] = Jc[i][r][d];
      for (int r = 0; r < 3; ++r) ltdl_solve(M, nv, pd, X[i][r]);
    }
    /* Delassus blocks G_ij = J_i M^-1 J_j^T and free contact velocity c_i = J_i u_free */
    double G[MAXK][MAXK][9], Ginv[MAXK][9], cfree[MAXK][3];
    for (int i = 0; i < nc; ++i) {
      for (int j = 0; j < nc; ++j)
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int d = 0; d < nv; ++d) s += Jc[i][r][d] * X[j][c][d];
            G[i][j][3 * r + c] = s;
          }
      if (lim_sign[i] != 0.0) G[i][i][0] = G[i][i][4] = 1.0;   /* dummy tangential diagonal of a joint-limit row */
      if (cbody2[i] >= 0 || m->fixed_base) {
        /* two bodies joined by fewer than three joints cannot move relative to each other in every direction (thigh against
         * trunk: two joints), so a self-collision's block can be rank deficient - and so can every block of a fixed-base
         * system (a body fewer than three joints from the world); a small compliance, relative to the block's mean diagonal,
         * keeps the per-contact rule well posed (once for either reason, twice for both) */
        const double reg = ((cbody2[i] >= 0) + (m->fixed_base != 0)) * ORC_SELF_REG * (G[i][i][0] + G[i][i][4] + G[i][i][8]) / 3.0;
        G[i][i][0] += reg; G[i][i][4] += reg; G[i][i][8] += reg;
      }
      inv3(G[i][i], Ginv[i]);
      for (int r = 0; r < 3; ++r) {
        double s = 0;
        for (int d = 0; d < nv; ++d) s += Jc[i][r][d] * ufree[d];
        cfree[i][r] = s;
      }
      cfree[i][2] -= p->erp * cdepth[i] / p->dt;
      const double e_i = cbody2[i] >= 0 ? cmat[i][1] : (i < nreal && p->col_restitution) ? p->col_restitution[ccol[i]] : p->restitution;
      const double thr_i = cbody2[i] >= 0 ? cmat[i][2] : (i < nreal && p->col_res_threshold) ? p->col_res_threshold[ccol[i]] : p->res_threshold;
      if (e_i > 0.0 && lim_sign[i] == 0.0) {   /* Newton restitution on the approach speed J u of this step */
        double vn0 = 0;
        for (int d = 0; d < nv; ++d) vn0 += Jc[i][2][d] * u[d];
        if (vn0 < -thr_i) cfree[i][2] += e_i * vn0;
      }
      /* warm start: the impulse (contact frame) this collision primitive carried in the previous integrate() */
      /* (a self-collision starts cold: the warm state is kept per primitive for its contact with the terrain) */
      for (int r = 0; r < 3; ++r) lam[i][r] = (lam_warm && p->warm_start && i < nreal && cbody2[i] < 0 && !csecond[i]) ? lam_warm[ORC_WARM * ccol[i] + r] : 0.0;
    }
    /* EXPERIMENT orc_params::reduce_dist: contact-set reduction.  The solver below then runs on the merged set (the arrays are compacted in place);
     * red_map / red_w / the saved per-contact arrays expand its result back to the original contacts behind the solve. */
    int red_on = 0, red_map[MAXK], nc0 = nc, nreal0 = nreal, cbody_s[MAXK], cbody2_s[MAXK], ccol_s[MAXK], csecond_s[MAXK];
    double red_w[MAXK], lim_s[MAXK];
    if (p->reduce_dist > 0.0 && nc > 1) {
      static _Thread_local double* tl_G2 = NULL;
      if (!tl_G2) tl_G2 = (double*)malloc(sizeof(double) * MAXK * MAXK * 9);
      int partner[MAXK], first[MAXK], nm = 0;
      for (int i = 0; i < nc; ++i) partner[i] = -1;
      for (int i = 0; i < nreal; ++i) {
        if (partner[i] >= 0 || cbody2[i] >= 0 || csecond[i] || lim_sign[i] != 0.0) continue;
        for (int j = i + 1; j < nreal; ++j) {
          if (partner[j] >= 0 || cbody2[j] >= 0 || csecond[j] || lim_sign[j] != 0.0 || cbody[j] != cbody[i]) continue;
          const double d[3] = {cx[i][0] - cx[j][0], cx[i][1] - cx[j][1], cx[i][2] - cx[j][2]};
          if (dot3(d, d) >= p->reduce_dist * p->reduce_dist || dot3(cn[i], cn[j]) < 1.0 - 1e-12) continue;
          partner[i] = j; partner[j] = i; red_on = 1;
          break;
        }
      }
      if (red_on) {
        for (int i = 0; i < nc; ++i) {          /* merged index of every original contact, its weight */
          if (partner[i] >= 0 && partner[i] < i) { red_map[i] = red_map[partner[i]]; }
          else { red_map[i] = nm; first[nm] = i; ++nm; }
          red_w[i] = 1.0;
          if (partner[i] >= 0) { const int j = partner[i]; red_w[i] = (cdepth[i] + 1e-4) / (cdepth[i] + cdepth[j] + 2e-4); }
        }
        /* G' = P^T G P, c' = P^T c, lam' = sum of the members' warm impulses; in place behind a copy of G */
        for (int a = 0; a < nc; ++a) for (int b = 0; b < nc; ++b) for (int e = 0; e < 9; ++e) tl_G2[(a * MAXK + b) * 9 + e] = G[a][b][e];
        double c2[MAXK][3], l2[MAXK][3];
        for (int a = 0; a < nm; ++a) { for (int r = 0; r < 3; ++r) { c2[a][r] = 0.0; l2[a][r] = 0.0; } for (int b = 0; b < nm; ++b) for (int e = 0; e < 9; ++e) G[a][b][e] = 0.0; }
        for (int i = 0; i < nc; ++i) {
          for (int r = 0; r < 3; ++r) { c2[red_map[i]][r] += red_w[i] * cfree[i][r]; l2[red_map[i]][r] += lam[i][r]; }
          for (int j = 0; j < nc; ++j) for (int e = 0; e < 9; ++e) G[red_map[i]][red_map[j]][e] += red_w[i] * red_w[j] * tl_G2[(i * MAXK + j) * 9 + e];
        }
        for (int a = 0; a < nc; ++a) { cbody_s[a] = cbody[a]; cbody2_s[a] = cbody2[a]; lim_s[a] = lim_sign[a]; ccol_s[a] = ccol[a]; csecond_s[a] = csecond[a]; }
        int nreal2 = 0;
        for (int a = 0; a < nm; ++a) {
          const int i = first[a];
          for (int r = 0; r < 3; ++r) { cfree[a][r] = c2[a][r]; lam[a][r] = l2[a][r]; }
          cbody[a] = cbody_s[i]; cbody2[a] = cbody2_s[i]; lim_sign[a] = lim_s[i]; ccol[a] = ccol_s[i]; csecond[a] = csecond_s[i];
          if (i < nreal) nreal2 = a + 1;
          if (lim_sign[a] != 0.0) G[a][a][0] = G[a][a][4] = 1.0;
          inv3(G[a][a], Ginv[a]);
        }
        nc = nm; nreal = nreal2;
      }
    }
    /* per-contact Gauss-Seidel (Hwangbo et al. 2018 Alg. 1) */
    if (dbgG)   /* debug views cover the real contacts (joint-limit rows follow them and are left out) */
      for (int i = 0; i < nreal; ++i)
        for (int j = 0; j < nreal; ++j)
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) dbgG[(3 * i + r) * 3 * nreal + 3 * j + c] = G[i][j][3 * r + c];
    if (dbgc) for (int i = 0; i < nreal; ++i) for (int r = 0; r < 3; ++r) dbgc[3 * i + r] = cfree[i][r];
    /* Convergence: largest impulse change of the sweep <= threshold * (largest normal impulse + floor).
     * A RELATIVE criterion on purpose: the device evaluates the same test in fp32, where the rounding
     * noise of an impulse update is proportional to the impulse magnitudes (RaiSim's absolute fp64
     * threshold [RECALL] cannot be met in fp32). */
    /* Stagnation exit: redundant contact sets (several contacts on one link of a crouched robot) make the
     * sweep cycle or crawl; the iteration is cut when the best relative error of the last `stall_window`
     * sweeps is not below `stall_factor` x the best of the window before.  Such solves would otherwise run to
     * max_iter without converging; on a lock-step GPU launch that worst case sets the launch time. */
    double cmu[MAXK];   /* friction coefficient of each contact = its collision primitive's material against the terrain */
    for (int i = 0; i < nc; ++i) cmu[i] = cbody2[i] >= 0 ? cmat[i][0] : (i < nreal && p->col_mu) ? p->col_mu[ccol[i]] : p->mu;
    double alpha = p->alpha_init, best_prev = 1e300, best_cur = 1e300;
    double sdir[MAXK][3], lam_best[MAXK][3], best_rel = 1e300;
    for (int i = 0; i < nc; ++i) {
      sdir[i][0] = sdir[i][1] = sdir[i][2] = 0.0;
      lam_best[i][0] = lam_best[i][1] = lam_best[i][2] = 0.0;
      if (lam_warm && p->warm_start && i < nreal && cbody2[i] < 0 && !csecond[i] && lam_warm[ORC_WARM * ccol[i] + 5] != 0.0) {   /* ... and its last friction direction */
        sdir[i][0] = lam_warm[ORC_WARM * ccol[i] + 3]; sdir[i][1] = lam_warm[ORC_WARM * ccol[i] + 4]; sdir[i][2] = 3.0;
      }
    }
    /* groups of the grouped sweep: limb = ancestor of the contact's body at level 1 (0 for the base); position of each
     * contact within its group (contact order), and the largest group = passes per sweep */
    int gid[MAXK], gpos[MAXK], gdepth = 0;
    for (int i = 0; i < nc; ++i) {
      int b = cbody[i];
      while (b > 0 && m->parent[b] > 0) b = m->parent[b];
      gid[i] = b;
    }
    if (nself > 0) {
      /* a self-collision couples its two limbs strongly: their groups are merged (relabelling in contact order) */
      int gid2[MAXK];
      for (int i = 0; i < nc; ++i) {
        int b = cbody2[i];
        while (b > 0 && m->parent[b] > 0) b = m->parent[b];
        gid2[i] = cbody2[i] >= 0 ? b : gid[i];
      }
      for (int j = 0; j < nc; ++j) {
        if (cbody2[j] < 0) continue;
        const int lo = gid[j] < gid2[j] ? gid[j] : gid2[j], hi = gid[j] < gid2[j] ? gid2[j] : gid[j];
        if (lo == hi) continue;
        for (int i = 0; i < nc; ++i) { if (gid[i] == hi) gid[i] = lo; if (gid2[i] == hi) gid2[i] = lo; }
      }
    }
    for (int i = 0; i < nc; ++i) {
      gpos[i] = 0;
      for (int j = 0; j < i; ++j) if (gid[j] == gid[i]) ++gpos[i];
      if (gpos[i] + 1 > gdepth) gdepth = gpos[i] + 1;
    }
    int converged = 0;
    /* multi-contact envs (>= multi_depth contacts on one limb: redundant sets) run with their own lag / stagnation settings */
    const int multi = p->multi_depth > 0 && gdepth >= p->multi_depth;
    const int freeze_after = multi ? p->multi_freeze_after : p->freeze_after;
    const int stall_window = multi ? p->multi_stall_window : p->stall_window;
    double aa_x[MAXK][3], aa_g[MAXK][3], aa_r[MAXK][3], aa_next[MAXK][3];
    int aa_have = 0, aa_apply = 0;
    const int aa_on = p->anderson > 0 && multi && p->kmax > 8;   /* the device carries it in its large-model kernel classes only */
    for (int it = 0; it < p->max_iter; ++it) {
      double err = 0, scale = 0;
      const int lag = freeze_after > 0 && it >= freeze_after;
      if (aa_on) for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) aa_x[i][r] = lam[i][r];
      if (p->group_parallel) {
        /* Grouped sweep (what the device runs).  Contacts are grouped by the limb they sit on: the subtree hanging off the
         * base that holds the contact's body; contacts on the base itself form one more group.  Contacts of DIFFERENT limbs
         * couple only through the base (|G_ij| ~ 0.1 |G_ii| on the quadruped); contacts of one limb - above all two contacts
         * on one link - couple strongly.  A sweep walks the k-th contact of every group AT ONCE (pass k: block Jacobi across
         * limbs - each member applies the per-contact rule to the impulses the pass started with) and the members of one group
         * in turn (Gauss-Seidel within a limb).  On the device a pass is ONE SIMD evaluation of the rule on all contact lanes,
         * where the sequential sweep needs one evaluation per contact.  The fixed points are those of the per-contact
         * iteration; on the benchmark population the sweep count is that of the sequential sweep + 6 % and the deviation from
         * the plain iteration is unchanged (tests/test_oracle_solver_heuristics.py). */
        /* Light passes (rounds 1-2; now opt-in, orc_params::multi_light): a multi-contact env refreshes the friction directions
         * of ALL its contacts in pass 0, from the impulses the sweep starts with; the later passes keep the directions and only
         * re-solve magnitudes (a contact without a usable direction runs the global search).  A third of the work per pass on
         * the device - and, measured in round 3 on the standing humanoid, 9 % of the solves unconverged after 150 sweeps where
         * refreshing inside every pass leaves 2 % (see orc_params).  group_parallel = 2 forces it for every env. */
        const int light = p->group_parallel == 2 || (multi && p->multi_light);
        /* group_parallel = 3 (ablation, "group-local sweep"): ONE snapshot per sweep - a member sees the new impulses of its own group's
         * earlier members and the sweep-start impulses of every other group (block Jacobi across limbs per SWEEP, not per pass).  Same
         * fixed points and sweep counts (humanoid 19.8 vs 19.4 / 16.8 vs 16.0, quadruped identical: tests/test_oracle_solver_heuristics.py).
         * Built on the device in round 3 for the kmax > 8 classes (one masked exchange per sweep + a ds_bpermute / LDS hand-over inside
         * the limb per pass) and measured SLOWER than the per-pass exchange (config 5 standing 10.9 vs 12.0 M): not what the device runs. */
        const int glocal = p->group_parallel == 3;
        /* Body-level stick solve (prototype, orc_params::body_stick).  Several contacts on ONE rigid body couple through the body's
         * 6 x 6 inverse operational inertia: G_ij = X_i L X_j^T with X_i = R_i^T [I | -[r_i]x] (contact-frame point velocity of a body
         * twist about the contacts' centroid) - a 3k x 3k block of rank <= 6 on which the per-contact sweeps crawl.  If all k contacts
         * stick, the body's contact points stop: with v0 = the point velocities WITHOUT the body's own impulses (a rigid twist t0,
         * v0_i = X_i t0), the net wrench is unique, w = -L^-1 t0, and lam_i = X_i (X^T X)^-1 w is the minimum-norm distribution that
         * produces it.  L = P (X^T G X) P with P = (X^T X)^-1 needs nothing but the blocks already there.  Accepted only when every
         * lam_i lies inside its cone (then it IS a solution of the per-contact conditions: zero velocity, admissible impulse);
         * otherwise the body's contacts take their ordinary passes.  The velocities it leads to are those of the per-contact
         * iteration (the stick solution is unique in u); the split of the wrench over the redundant contacts is not unique anyway. */
        int body_done[MAXK];
        for (int i = 0; i < nc; ++i) body_done[i] = 0;
        if (p->body_stick) {
          for (int i0 = 0; i0 < nc; ++i0) {
            if (body_done[i0] || cbody2[i0] >= 0 || lim_sign[i0] != 0.0 || i0 >= nreal) continue;
            int idx[MAXK], k = 0;
            for (int j = i0; j < nreal; ++j) if (cbody[j] == cbody[i0] && cbody2[j] < 0 && lim_sign[j] == 0.0 && !body_done[j]) idx[k++] = j;
            if (k < 2) continue;
            if (k == 2) {
              /* two contacts on one body (a foot on its edge, a shank on knee + foot): the stick equations G_BB lam = -v0 (6 x 6) have
               * rank 5 - the squeeze mode n = (+d, -d) along the line through the two points produces no wrench.  Solve with that known
               * null vector deflated, (G_BB + s n n^T) lam = -v0, and project it out again: the minimum-norm all-stick impulses. */
              const int ia = idx[0], ib = idx[1];
              double d[3] = {cx[ib][0] - cx[ia][0], cx[ib][1] - cx[ia][1], cx[ib][2] - cx[ia][2]};
              const double dn = sqrt(dot3(d, d));
              if (!(dn > 1e-9)) continue;
              for (int c = 0; c < 3; ++c) d[c] /= dn;
              double n6[6], A[36], Ai[36], rhs[6], l6[6];
              for (int ax = 0; ax < 3; ++ax) {
                n6[ax] = (Rc[ia][ax] * d[0] + Rc[ia][3 + ax] * d[1] + Rc[ia][6 + ax] * d[2]) * 0.70710678118654752;
                n6[3 + ax] = -(Rc[ib][ax] * d[0] + Rc[ib][3 + ax] * d[1] + Rc[ib][6 + ax] * d[2]) * 0.70710678118654752;
              }
              double tr = 0;
              for (int a = 0; a < 2; ++a) for (int b2 = 0; b2 < 2; ++b2) for (int ra = 0; ra < 3; ++ra) for (int rb = 0; rb < 3; ++rb)
                A[6 * (3 * a + ra) + 3 * b2 + rb] = G[idx[a]][idx[b2]][3 * ra + rb];
              for (int c = 0; c < 6; ++c) tr += A[7 * c];
              for (int c = 0; c < 6; ++c) for (int e2 = 0; e2 < 6; ++e2) A[6 * c + e2] += (tr / 6.0) * n6[c] * n6[e2];
              if (!inv6(A, Ai, 1e-10)) continue;
              for (int a = 0; a < 2; ++a) {
                const int i = idx[a];
                double v0[3] = {cfree[i][0], cfree[i][1], cfree[i][2]};
                for (int j = 0; j < nc; ++j) {
                  if (j == ia || j == ib) continue;
                  for (int r2 = 0; r2 < 3; ++r2) v0[r2] += G[i][j][3 * r2] * lam[j][0] + G[i][j][3 * r2 + 1] * lam[j][1] + G[i][j][3 * r2 + 2] * lam[j][2];
                }
                for (int r2 = 0; r2 < 3; ++r2) rhs[3 * a + r2] = -v0[r2];
              }
              double nl = 0;
              for (int c = 0; c < 6; ++c) { double t = 0; for (int e2 = 0; e2 < 6; ++e2) t += Ai[6 * c + e2] * rhs[e2]; l6[c] = t; }
              for (int c = 0; c < 6; ++c) nl += n6[c] * l6[c];
              for (int c = 0; c < 6; ++c) l6[c] -= nl * n6[c];
              int ok2 = 1;
              for (int a = 0; a < 2 && ok2; ++a) {
                const double mu_i = cmu[idx[a]];
                ok2 = l6[3 * a + 2] >= 0.0 && l6[3 * a] * l6[3 * a] + l6[3 * a + 1] * l6[3 * a + 1] <= mu_i * mu_i * l6[3 * a + 2] * l6[3 * a + 2];
              }
              if (!ok2) continue;
              for (int a = 0; a < 2; ++a) {
                const int i = idx[a];
                for (int r2 = 0; r2 < 3; ++r2) { const double dl = l6[3 * a + r2] - lam[i][r2]; lam[i][r2] = l6[3 * a + r2]; if (fabs(dl) > err) err = fabs(dl); }
                body_done[i] = 1;
              }
              continue;
            }
            double cen[3] = {0, 0, 0};
            for (int a = 0; a < k; ++a) for (int c = 0; c < 3; ++c) cen[c] += cx[idx[a]][c] / k;
            double Xb[MAXK][3][6];     /* X_i: rows = contact axes (t1, t2, n), columns = twist (v_ref, omega) */
            for (int a = 0; a < k; ++a) {
              const int i = idx[a];
              const double r[3] = {cx[i][0] - cen[0], cx[i][1] - cen[1], cx[i][2] - cen[2]};
              for (int ax = 0; ax < 3; ++ax) {
                const double e[3] = {Rc[i][ax], Rc[i][3 + ax], Rc[i][6 + ax]};      /* column ax of Rc = the axis in world coordinates */
                double rxe[3];
                cross3(r, e, rxe);                                                 /* e . (omega x r) = omega . (r x e) */
                for (int c = 0; c < 3; ++c) { Xb[a][ax][c] = e[c]; Xb[a][ax][3 + c] = rxe[c]; }
              }
            }
            double XtX[36], P[36];
            for (int q = 0; q < 36; ++q) XtX[q] = 0;
            for (int a = 0; a < k; ++a) for (int ax = 0; ax < 3; ++ax) for (int c = 0; c < 6; ++c) for (int d = 0; d < 6; ++d) XtX[6 * c + d] += Xb[a][ax][c] * Xb[a][ax][d];
            if (!inv6(XtX, P, 1e-9)) continue;                                      /* collinear contact points: the body can still turn about their line */
            double XtGX[36], T1[36], Li[36], Lm[36];
            for (int q = 0; q < 36; ++q) XtGX[q] = 0;
            for (int a = 0; a < k; ++a) for (int b2 = 0; b2 < k; ++b2) {
              const double* Gab = G[idx[a]][idx[b2]];
              for (int ra = 0; ra < 3; ++ra) for (int rb = 0; rb < 3; ++rb) {
                const double g = Gab[3 * ra + rb];
                for (int c = 0; c < 6; ++c) for (int d = 0; d < 6; ++d) XtGX[6 * c + d] += Xb[a][ra][c] * g * Xb[b2][rb][d];
              }
            }
            for (int c = 0; c < 6; ++c) for (int d = 0; d < 6; ++d) { double t = 0; for (int e2 = 0; e2 < 6; ++e2) t += P[6 * c + e2] * XtGX[6 * e2 + d]; T1[6 * c + d] = t; }
            for (int c = 0; c < 6; ++c) for (int d = 0; d < 6; ++d) { double t = 0; for (int e2 = 0; e2 < 6; ++e2) t += T1[6 * c + e2] * P[6 * e2 + d]; Li[6 * c + d] = t; }
            if (!inv6(Li, Lm, 1e-10)) continue;
            /* point velocities without the body's own impulses, as a twist */
            double xtv[6] = {0, 0, 0, 0, 0, 0};
            for (int a = 0; a < k; ++a) {
              const int i = idx[a];
              double v0[3] = {cfree[i][0], cfree[i][1], cfree[i][2]};
              for (int j = 0; j < nc; ++j) {
                int own = 0;
                for (int b2 = 0; b2 < k; ++b2) own |= idx[b2] == j;
                if (own) continue;
                for (int r2 = 0; r2 < 3; ++r2) v0[r2] += G[i][j][3 * r2] * lam[j][0] + G[i][j][3 * r2 + 1] * lam[j][1] + G[i][j][3 * r2 + 2] * lam[j][2];
              }
              for (int ax = 0; ax < 3; ++ax) for (int c = 0; c < 6; ++c) xtv[c] += Xb[a][ax][c] * v0[ax];
            }
            double t0[6], w[6], pw[6];
            for (int c = 0; c < 6; ++c) { double t = 0; for (int d = 0; d < 6; ++d) t += P[6 * c + d] * xtv[d]; t0[c] = t; }
            for (int c = 0; c < 6; ++c) { double t = 0; for (int d = 0; d < 6; ++d) t += Lm[6 * c + d] * t0[d]; w[c] = -t; }
            for (int c = 0; c < 6; ++c) { double t = 0; for (int d = 0; d < 6; ++d) t += P[6 * c + d] * w[d]; pw[c] = t; }
            double lc[MAXK][3];
            int ok = 1;
            for (int a = 0; a < k && ok; ++a) {
              for (int ax = 0; ax < 3; ++ax) { double t = 0; for (int c = 0; c < 6; ++c) t += Xb[a][ax][c] * pw[c]; lc[a][ax] = t; }
              const double mu_i = cmu[idx[a]];
              ok = lc[a][2] >= 0.0 && lc[a][0] * lc[a][0] + lc[a][1] * lc[a][1] <= mu_i * mu_i * lc[a][2] * lc[a][2];
            }
            if (!ok) continue;
            for (int a = 0; a < k; ++a) {
              const int i = idx[a];
              for (int r2 = 0; r2 < 3; ++r2) { const double dl = lc[a][r2] - lam[i][r2]; lam[i][r2] = lc[a][r2]; if (fabs(dl) > err) err = fabs(dl); }
              body_done[i] = 1;
            }
          }
        }
        double lamS[MAXK][3];
        int pair_done[MAXK];
        for (int i = 0; i < nc; ++i) { pair_done[i] = 0; for (int r = 0; r < 3; ++r) lamS[i][r] = lam[i][r]; }
        for (int kpos = 0; kpos < gdepth; ++kpos) {
          double lam0[MAXK][3];
          for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) lam0[i][r] = lam[i][r];
          if (glocal)   /* other groups: as the sweep started; own group: current (handled per member below) */
            for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) lam0[i][r] = lamS[i][r];
          for (int i = 0; i < nc; ++i) {
            if (gpos[i] != kpos && !(light && kpos == 0)) continue;
            if (body_done[i]) continue;     /* its body took the all-stick solution in this sweep */
            if (pair_done[i]) continue;     /* EXPERIMENT orc_params::pair_inner: solved together with its partner on the same link */
            if (p->pair_inner > 0 && !light && gpos[i] == kpos && i < nreal && cbody2[i] < 0) {
              int j = -1;
              for (int c2 = i + 1; c2 < nreal; ++c2)
                if (gid[c2] == gid[i] && gpos[c2] == kpos + 1 && cbody[c2] == cbody[i] && cbody2[c2] < 0 && !body_done[c2]) { j = c2; break; }
              if (j >= 0) {
                double vi0[3] = {cfree[i][0], cfree[i][1], cfree[i][2]}, vj0[3] = {cfree[j][0], cfree[j][1], cfree[j][2]};
                for (int m2 = 0; m2 < nc; ++m2) {
                  if (m2 == i || m2 == j) continue;
                  for (int r = 0; r < 3; ++r) {
                    vi0[r] += G[i][m2][3 * r] * lam0[m2][0] + G[i][m2][3 * r + 1] * lam0[m2][1] + G[i][m2][3 * r + 2] * lam0[m2][2];
                    vj0[r] += G[j][m2][3 * r] * lam0[m2][0] + G[j][m2][3 * r + 1] * lam0[m2][1] + G[j][m2][3 * r + 2] * lam0[m2][2];
                  }
                }
                double li[3] = {lam0[i][0], lam0[i][1], lam0[i][2]}, lj[3] = {lam0[j][0], lam0[j][1], lam0[j][2]};
                for (int t = 0; t < p->pair_inner; ++t) {
                  double v[3], ln[3], ch = 0.0;
                  for (int r = 0; r < 3; ++r) v[r] = vi0[r] + G[i][j][3 * r] * lj[0] + G[i][j][3 * r + 1] * lj[1] + G[i][j][3 * r + 2] * lj[2];
                  solve_one_contact(G[i][i], Ginv[i], v, cmu[i], p->section_rounds, lag, p->refine, 0.0, sdir[i], ln, p->slip_rule);
                  for (int r = 0; r < 3; ++r) { const double d = alpha * (ln[r] - li[r]); li[r] += d; if (fabs(d) > ch) ch = fabs(d); }
                  for (int r = 0; r < 3; ++r) v[r] = vj0[r] + G[j][i][3 * r] * li[0] + G[j][i][3 * r + 1] * li[1] + G[j][i][3 * r + 2] * li[2];
                  solve_one_contact(G[j][j], Ginv[j], v, cmu[j], p->section_rounds, lag, p->refine, 0.0, sdir[j], ln, p->slip_rule);
                  for (int r = 0; r < 3; ++r) { const double d = alpha * (ln[r] - lj[r]); lj[r] += d; if (fabs(d) > ch) ch = fabs(d); }
                  orc_pair_evals += 2;
                  double sc = li[2] > lj[2] ? li[2] : lj[2];
                  if (ch <= 1e-3 * p->threshold * (sc + ORC_LAMBDA_FLOOR)) break;
                }
                for (int r = 0; r < 3; ++r) {
                  const double di = li[r] - lam0[i][r], dj = lj[r] - lam0[j][r];
                  lam[i][r] = li[r]; lam[j][r] = lj[r];
                  if (fabs(di) > err) err = fabs(di);
                  if (fabs(dj) > err) err = fabs(dj);
                }
                pair_done[j] = 1;
                continue;
              }
            }
            double v[3] = {cfree[i][0], cfree[i][1], cfree[i][2]}, ln[3];
            for (int j = 0; j < nc; ++j) {
              if (j == i) continue;
              const double* lj = (glocal && gid[j] == gid[i]) ? lam[j] : lam0[j];
              for (int r = 0; r < 3; ++r) v[r] += G[i][j][3 * r] * lj[0] + G[i][j][3 * r + 1] * lj[1] + G[i][j][3 * r + 2] * lj[2];
            }
            if (light && kpos > 0) solve_one_contact(G[i][i], Ginv[i], v, cmu[i], p->section_rounds, 1, 0, 0.0, sdir[i], ln, p->slip_rule);
            else solve_one_contact(G[i][i], Ginv[i], v, cmu[i], p->section_rounds, lag, p->refine, 0.0, sdir[i], ln, p->slip_rule);
            if (gpos[i] != kpos) continue;   /* light variant, pass 0: a later member only refreshed its direction */
            for (int r = 0; r < 3; ++r) {
              const double base = glocal ? lam[i][r] : lam0[i][r];
              double dl = alpha * (ln[r] - base);
              lam[i][r] = base + dl;
              if (fabs(dl) > err) err = fabs(dl);
            }
          }
        }
        /* an inherited direction that the first sweep did not pick up is dropped: a contact that starts to slip later in the
         * solve runs the global search */
        if (it == 0) for (int i = 0; i < nc; ++i) if (sdir[i][2] == 3.0) sdir[i][2] = 0.0;
      } else {
      if (p->dir_per_sweep) {
        /* friction directions are refreshed ONCE per sweep, for all contacts from the impulses the sweep starts with
         * (on the device: one SIMD pass over the contact lanes instead of a refinement inside every sequential contact
         * update); the Gauss-Seidel pass below then keeps them fixed and only re-solves magnitudes */
        double lam0[MAXK][3], tmp[3];
        for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) lam0[i][r] = lam[i][r];
        for (int i = 0; i < nc; ++i) {
          double v[3] = {cfree[i][0], cfree[i][1], cfree[i][2]};
          for (int j = 0; j < nc; ++j) {
            if (j == i) continue;
            for (int r = 0; r < 3; ++r) v[r] += G[i][j][3 * r] * lam0[j][0] + G[i][j][3 * r + 1] * lam0[j][1] + G[i][j][3 * r + 2] * lam0[j][2];
          }
          solve_one_contact(G[i][i], Ginv[i], v, cmu[i], p->section_rounds, lag, p->refine, 0.0, sdir[i], tmp, p->slip_rule);
        }
        /* an inherited direction that the first refresh did not pick up is dropped: a contact that starts to slip later in
         * the solve runs the global search (the device checks inherited directions against the coarse scan in the first
         * sweep only) */
        if (it == 0) for (int i = 0; i < nc; ++i) if (sdir[i][2] == 3.0) sdir[i][2] = 0.0;
      }
      for (int i = 0; i < nc; ++i) {
        double v[3] = {cfree[i][0], cfree[i][1], cfree[i][2]}, ln[3];
        for (int j = 0; j < nc; ++j) {
          if (j == i) continue;
          for (int r = 0; r < 3; ++r) v[r] += G[i][j][3 * r] * lam[j][0] + G[i][j][3 * r + 1] * lam[j][1] + G[i][j][3 * r + 2] * lam[j][2];
        }
        /* per-sweep mode: the pass keeps every usable direction (frozen formula); a contact without one - it started to slip
         * inside this sweep, or its direction is inherited / ill conditioned - runs the global search right here (no Newton) */
        solve_one_contact(G[i][i], Ginv[i], v, cmu[i], p->section_rounds,
                          p->dir_per_sweep ? 1 : lag, p->dir_per_sweep ? 0 : p->refine, p->dir_per_sweep ? 0.0 : p->settle_tol, sdir[i], ln, p->slip_rule);
        for (int r = 0; r < 3; ++r) {
          double dl = alpha * (ln[r] - lam[i][r]);
          lam[i][r] += dl;
          if (fabs(dl) > err) err = fabs(dl);
        }
      }
      }   /* sequential sweeps (ablations) */
      if (aa_on) {
        /* Anderson acceleration, depth 1 (orc_params::anderson).  A sweep is a fixed-point map g; with x the impulses the
         * sweep started from, r = g(x) - x.  From two consecutive pairs the secant step x+ = g - gamma (g - g_prev),
         * gamma = <r, r - r_prev> / |r - r_prev|^2, is exact for an affine contraction with one dominant mode - which is what the
         * crawl of redundant sticking contact sets is - and is projected back into the friction cones.  The convergence test stays
         * the sweep's own |g(x) - x|: an extrapolated iterate is only ever the START of a sweep, never returned unchecked. */
        double rr[MAXK][3], num = 0, den = 0;
        for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) rr[i][r] = lam[i][r] - aa_x[i][r];
        if (aa_have && it + 1 >= p->anderson) {
          for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) { const double dr = rr[i][r] - aa_r[i][r]; num += rr[i][r] * dr; den += dr * dr; }
        }
        double gam = den > 1e-300 ? num / den : 0.0;
        if (!(fabs(gam) <= p->anderson_clip)) gam = 0.0;
        for (int i = 0; i < nc; ++i) {
          double xn[3];
          for (int r = 0; r < 3; ++r) { xn[r] = lam[i][r] - gam * (lam[i][r] - aa_g[i][r]); aa_g[i][r] = lam[i][r]; aa_r[i][r] = rr[i][r]; }
          if (xn[2] <= 0.0) xn[0] = xn[1] = xn[2] = 0.0;
          else {
            const double t = sqrt(xn[0] * xn[0] + xn[1] * xn[1]), lim = cmu[i] * xn[2];
            if (t > lim) { xn[0] *= lim / t; xn[1] *= lim / t; }
          }
          for (int r = 0; r < 3; ++r) aa_next[i][r] = xn[r];
        }
        aa_have = 1; aa_apply = gam != 0.0;
      }
      for (int i = 0; i < nc; ++i) if (lam[i][2] > scale) scale = lam[i][2];
      it_used = it + 1;
      alpha = alpha * p->alpha_decay;
      if (alpha < p->alpha_min) alpha = p->alpha_min;
      if (err <= p->threshold * (scale + ORC_LAMBDA_FLOOR)) { converged = 1; break; }
      double rel = err / (scale + ORC_LAMBDA_FLOOR);
      if (rel < best_cur) best_cur = rel;
      if (rel < best_rel) {  /* remember the calmest iterate: a solve that does not converge returns it */
        best_rel = rel;
        for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) lam_best[i][r] = lam[i][r];
      }
      if (stall_window > 0 && (it + 1) % stall_window == 0) {
        if (best_cur > p->stall_factor * best_prev) break;
        best_prev = best_cur; best_cur = 1e300;
      }
      if (aa_on && aa_apply) for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) lam[i][r] = aa_next[i][r];
    }
    if (!converged) {
      /* per-contact iteration that cycles or crawls can sit at a wild iterate when it is cut off (measured: a
       * 1.9e4 N s impulse on a jammed shank); return the iterate with the smallest sweep-to-sweep change instead */
      fl |= 4;
      for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) lam[i][r] = lam_best[i][r];
    }
    if (red_on) {      /* back to the original contacts: lam_i = w_i lam_m, the merged contact's friction direction for both members */
      double lm[MAXK][3], sm[MAXK][3];
      for (int a = 0; a < nc; ++a) for (int r = 0; r < 3; ++r) { lm[a][r] = lam[a][r]; sm[a][r] = sdir[a][r]; }
      nc = nc0; nreal = nreal0;
      for (int i = 0; i < nc; ++i) {
        for (int r = 0; r < 3; ++r) { lam[i][r] = red_w[i] * lm[red_map[i]][r]; sdir[i][r] = sm[red_map[i]][r]; }
        cbody[i] = cbody_s[i]; cbody2[i] = cbody2_s[i]; lim_sign[i] = lim_s[i]; ccol[i] = ccol_s[i]; csecond[i] = csecond_s[i];
      }
    }
    if (dbglam) for (int i = 0; i < nreal; ++i) for (int r = 0; r < 3; ++r) dbglam[3 * i + r] = lam[i][r];
    for (int i = 0; i < nc; ++i) for (int r = 0; r < 3; ++r) sdir_out[i][r] = sdir[i][r];
  }

  /* u+ = u_free + M^-1 J^T lam ;  q+ = q (+) dt u+   (semi-implicit Euler) */
  for (int i = 0; i < nc; ++i)
    for (int r = 0; r < 3; ++r)
      for (int d = 0; d < nv; ++d) ufree[d] += X[i][r][d] * lam[i][r];
  /* the position update's velocity: theta u+ + (1 - theta) u  (orc_params::integ_theta: 1 = semi-implicit Euler, RaiSim's default; 0 = explicit
   * Euler; 0.5 = trapezoid) */
  double ub[MAXV];
  for (int d = 0; d < nv; ++d) { ub[d] = p->integ_theta * ufree[d] + (1.0 - p->integ_theta) * u[d]; u[d] = ufree[d]; }
  for (int c = 0; c < 3; ++c) q[c] += p->dt * ub[c];
  {
    double w[3] = {ub[3], ub[4], ub[5]};
    double wn = sqrt(dot3(w, w)), half = 0.5 * wn * p->dt;
    double sc = (wn > 1e-12) ? sin(half) / wn : 0.5 * p->dt, cw = cos(half);
    double dq[4] = {cw, sc * w[0], sc * w[1], sc * w[2]};
    double a[4] = {q[3], q[4], q[5], q[6]}, r4[4];
    r4[0] = dq[0] * a[0] - dq[1] * a[1] - dq[2] * a[2] - dq[3] * a[3];
    r4[1] = dq[0] * a[1] + dq[1] * a[0] + dq[2] * a[3] - dq[3] * a[2];
    r4[2] = dq[0] * a[2] - dq[1] * a[3] + dq[2] * a[0] + dq[3] * a[1];
    r4[3] = dq[0] * a[3] + dq[1] * a[2] - dq[2] * a[1] + dq[3] * a[0];
    double n4 = 1.0 / sqrt(r4[0] * r4[0] + r4[1] * r4[1] + r4[2] * r4[2] + r4[3] * r4[3]);
    for (int c = 0; c < 4; ++c) q[3 + c] = r4[c] * n4;
  }
  for (int i = 1; i < m->nb; ++i) q[qidx_of(i)] += p->dt * ub[dof_of(i)];

  for (int i = 0; i < nq; ++i) if (!isfinite(q[i])) fl |= 2;
  for (int i = 0; i < nv; ++i) if (!isfinite(u[i])) fl |= 2;

  if (lam_warm) {
    for (int i = 0; i < ORC_WARM * m->ncol; ++i) lam_warm[i] = 0.0;
    for (int i = 0; i < nreal; ++i) {
      if (cbody2[i] >= 0 || csecond[i]) continue;
      double* wrm = lam_warm + ORC_WARM * ccol[i];
      for (int r = 0; r < 3; ++r) wrm[r] = lam[i][r];
      if (sdir_out[i][2] != 0.0) { wrm[3] = sdir_out[i][0]; wrm[4] = sdir_out[i][1]; wrm[5] = 1.0; }
    }
  }
  int nout = 0;
  for (int i = 0; i < nreal; ++i) {
    const int two = cbody2[i] >= 0;
    for (int h = 0; h <= two; ++h, ++nout) {
      if (!contacts) continue;
      const double sg = h ? -1.0 : 1.0;
      for (int c = 0; c < 3; ++c) {
        contacts[nout].position[c] = k->pbase[c] + cx[i][c];
        contacts[nout].normal[c] = sg * cn[i][c];
        contacts[nout].impulse[c] = sg * (Rc[i][3 * c] * lam[i][0] + Rc[i][3 * c + 1] * lam[i][1] + Rc[i][3 * c + 2] * lam[i][2]);
      }
      contacts[nout].depth = cdepth[i];
      contacts[nout].body = h ? cbody2[i] : cbody[i];
      contacts[nout].collision = two ? (h ? (ccol2[i] | ORC_SELF_B) : (ccol[i] | ORC_SELF_A)) : (csecond[i] == 2 ? (ccol[i] | ORC_CAPSULE) : csecond[i] ? (ccol[i] | ORC_SECOND) : ccol[i]);
    }
  }
  if (n_contacts) *n_contacts = nout;
  if (iters) *iters = it_used;
  if (flags) *flags = fl;
}

void orc_step(const rsb_model_blob* m, const orc_params* p, double* q, double* u, const double* kp,
              const double* kd, const double* p_target, const double* d_target,
              const double* tau_ff, orc_contact* contacts, int32_t* n_contacts, int32_t* iters,
              int32_t* flags) {
  step_impl(m, p, q, u, kp, kd, p_target, d_target, tau_ff, contacts, n_contacts, iters, flags, NULL, NULL, NULL, NULL);
}

void orc_step_warm(const rsb_model_blob* m, const orc_params* p, double* q, double* u, const double* kp,
                   const double* kd, const double* p_target, const double* d_target,
                   const double* tau_ff, orc_contact* contacts, int32_t* n_contacts, int32_t* iters,
                   int32_t* flags, double* lam_warm) {
  step_impl(m, p, q, u, kp, kd, p_target, d_target, tau_ff, contacts, n_contacts, iters, flags, lam_warm, NULL, NULL, NULL);
}

void orc_step_debug(const rsb_model_blob* m, const orc_params* p, double* q, double* u, const double* kp,
                    const double* kd, const double* p_target, const double* d_target,
                    const double* tau_ff, orc_contact* contacts, int32_t* n_contacts, int32_t* iters,
                    int32_t* flags, double* lam_warm, double* G, double* c, double* lam) {
  step_impl(m, p, q, u, kp, kd, p_target, d_target, tau_ff, contacts, n_contacts, iters, flags, lam_warm, G, c, lam);
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int orc_step_batch(const rsb_model_blob* m, const orc_params* p, int N, int substeps, double* q,
                   double* u, const double* kp, const double* kd, const double* p_target,
                   const double* d_target, const double* tau_ff, orc_contact* contacts,
                   int32_t* n_contacts, int32_t* iters, int32_t* flags, double* lam_warm, int nthreads) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  used = nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
  for (int e = 0; e < N; ++e) {
    int fl_acc = 0;
    orc_params pe = *p;   /* this env's terrain: its own map of a curriculum (hm_index), else the shared one */
    if (p->hm_index && p->terrain_type == 1) pe.hm_heights = p->hm_heights + (size_t)p->hm_index[e] * p->hm_xs * p->hm_ys;
    for (int s = 0; s < substeps; ++s) {
      int32_t fl = 0;
      orc_step_warm(m, &pe, q + (size_t)e * m->nq, u + (size_t)e * m->nv, kp, kd,
                    p_target ? p_target + (size_t)e * m->nq : NULL,
                    d_target ? d_target + (size_t)e * m->nv : NULL,
                    tau_ff ? tau_ff + (size_t)e * m->nv : NULL,
                    contacts ? contacts + (size_t)e * p->kmax : NULL,
                    n_contacts ? n_contacts + e : NULL, iters ? iters + e : NULL, &fl,
                    lam_warm ? lam_warm + (size_t)e * ORC_WARM * m->ncol : NULL);
      fl_acc |= fl;
    }
    if (flags) flags[e] = fl_acc;
  }
  return used;
}
