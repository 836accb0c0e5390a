// dev_factors.h — device-side factor arithmetic (gfx950, FP64).
//
// Visual factor: ProjectionTdFactor / ProjectionFactor::Evaluate
// (vins_estimator/src/factor/projection_td_factor.cpp:36-151, projection_factor.cpp:21-121)
// re-derived around per-frame-pair uniform matrices (struct Tab) so that an observation
// costs ~250 FLOP instead of ~1k:  X_cj = T_ij X_ci + c_ij, and every translation Jacobian
// is the 2x3 "reduce" matrix times a pair-uniform 3x3:
//     dr/dPi = red M1_j        dr/dPj = -red M1_j        dr/dtic = red (M2_ij - ric^T)
// so only 14 "basis" columns [red(3) | dth_i(3) | dth_j(3) | dth_ic(3) | td | r] are formed per
// observation; the 20-column factor Jacobian is basis * E_ij with E_ij pair-uniform.
// IMU factor: IMUFactor::Evaluate (factor/imu_factor.h:19-200) and
// IntegrationBase::evaluate (factor/integration_base.h:160-186), literal.
#pragma once
#include "dev_math.h"
#include "dev_types.h"

struct ObsPair {
  d3 pi, vi, pj, vj;
  double tdi, tdj, rowi, rowj;
};

struct PairU {  // pair-uniform inputs
  m33 M2, T, ric, ricT;
  d3 c, tic;
  // Non-null for a pair whose frame j or extrinsic quaternion is off the unit sphere (dev_types.h, struct Tab): T and c above are
  // the residual chain's then, and the Jacobians fetch what they need beyond M2 from the table themselves (a start point's
  // newest frame — the lanes that go there are few, and nothing of it is live on the way everybody else takes)
  const double *offc;  // the pair's Jacobian-flavour c, Tab::c[tab_cj(pair)]
  const double *offr;  // ricF, Tab::T[0]
};
DEV int pair_is_off(unsigned offm, int j) { return (offm & ((1u << j) | (1u << TAB_EX_BIT))) != 0; }
DEV unsigned tab_offmask(const Tab *T) { return (unsigned)uniform_cptr(&T->c[0][0])[0]; }  // (written by an earlier kernel: a scalar load)

struct Basis {
  d3 red[2];   // reduce rows (corrected)
  d3 jti[2];   // d r / d theta_i
  d3 jtj[2];   // d r / d theta_j
  d3 jtx[2];   // d r / d theta_ic
  double jl[2];
  double jtd[2];
  double r[2];
  double rho0;
};

// Loads the 8 SoA channels of observation o.
DEV void load_obs(const Slot *S, int o, d3 &p, d3 &v, double &td, double &row) {
  p = mk3(S->obs[0][o], S->obs[1][o], S->obs[2][o]);
  v = mk3(S->obs[3][o], S->obs[4][o], S->obs[5][o]);
  td = S->obs[6][o];
  row = S->obs[7][o];
}

// residual only (cost sweep).  Returns rho(s) = log(1 + s) (ceres::CauchyLoss(1.0)).
DEV double visual_cost(const ObsPair &ob, double lam, double td, int est_td, double tr_over_row, double half_row,
                       double sqrt_info, const m33 &T, d3 c) {
  d3 pi = ob.pi, pj = ob.pj;
  if (est_td) {
    pi = ob.pi - (td - ob.tdi + tr_over_row * (ob.rowi - half_row)) * ob.vi;  // projection_td_factor.cpp:54
    pj = ob.pj - (td - ob.tdj + tr_over_row * (ob.rowj - half_row)) * ob.vj;  // :55
  }
  d3 Xci = (1.0 / lam) * pi;
  d3 Xcj = mul(T, Xci) + c;
  d3 nh = rsqrt(dot(Xcj, Xcj)) * Xcj;
  d3 pjn = rsqrt(dot(pj, pj)) * pj;
  // tangent basis from the UN-shifted pts_j (projection_td_factor.cpp:23-33)
  d3 a = rsqrt(dot(ob.pj, ob.pj)) * ob.pj;
  d3 tmp = mk3(0, 0, 1);
  if (a.x == 0.0 && a.y == 0.0 && a.z == 1.0) tmp = mk3(1, 0, 0);
  d3 b1 = tmp - dot(a, tmp) * a;
  b1 = rsqrt(dot(b1, b1)) * b1;
  d3 b2 = cross(a, b1);
  d3 d = nh - pjn;
  double r0 = sqrt_info * dot(b1, d), r1 = sqrt_info * dot(b2, d);
  return log(1.0 + (r0 * r0 + r1 * r1));
}

// residual + basis Jacobian columns, robust-corrected (ceres Corrector with rho'' <= 0:
// everything scaled by sqrt(rho'), marginalization_factor.cpp:49-53).
// OFFS = false: compiled without the off-sphere flavour (u.offc is not looked at) — the caller has seen a zero mask.
template <bool OFFS = true>
DEV void visual_basis(const ObsPair &ob, double lam, double td, int est_td, double tr_over_row, double half_row,
                      double sqrt_info, const PairU &u, Basis &B) {
  d3 pi = ob.pi, pj = ob.pj;
  if (est_td) {
    pi = ob.pi - (td - ob.tdi + tr_over_row * (ob.rowi - half_row)) * ob.vi;
    pj = ob.pj - (td - ob.tdj + tr_over_row * (ob.rowj - half_row)) * ob.vj;
  }
  const double inv_lam = 1.0 / lam;
  d3 Xci = inv_lam * pi;                 // pts_camera_i           (:56)
  d3 Xbi = mul(u.ric, Xci) + u.tic;      // pts_imu_i              (:57)
  d3 Xcj = mul(u.T, Xci) + u.c;          // pts_camera_j           (:58-60 folded)
  d3 Xbj = mul(u.ric, Xcj) + u.tic;      // pts_imu_j
  // The Jacobians write the skew argument of :131 with transposes where the chain above went through Quaternion::inverse() —
  // the same unless a quaternion of the pair is off the unit sphere:  (M2 ric) Xci + c = M2 (Xbi - tic) + c  needs no second T.
  d3 XcjJ = Xcj;
  if (OFFS && u.offc) {
    Xbj = mul(ldm(u.offr), Xcj) + u.tic;  // the chain's pts_imu_j (:59): ricF undoes R(qic^-1) exactly
    XcjJ = mul(u.M2, Xbi - u.tic) + ld3(u.offc);
  }
  const double inv_n = rsqrt(dot(Xcj, Xcj));
  d3 nh = inv_n * Xcj;
  d3 pjn = rsqrt(dot(pj, pj)) * pj;
  d3 a = rsqrt(dot(ob.pj, ob.pj)) * ob.pj;
  d3 tmp = mk3(0, 0, 1);
  if (a.x == 0.0 && a.y == 0.0 && a.z == 1.0) tmp = mk3(1, 0, 0);
  d3 b1 = tmp - dot(a, tmp) * a;
  b1 = rsqrt(dot(b1, b1)) * b1;
  d3 b2 = cross(a, b1);
  d3 d = nh - pjn;
  double r0 = sqrt_info * dot(b1, d), r1 = sqrt_info * dot(b2, d);  // :69,:75
  // reduce = sqrt_info * tangent_base * (I/n - X X^T / n^3)                      (:84-99)
  const double sn = sqrt_info * inv_n;
  d3 red0 = sn * (b1 - dot(b1, nh) * nh);
  d3 red1 = sn * (b2 - dot(b2, nh) * nh);
  // robust weight
  const double s2 = r0 * r0 + r1 * r1;
  const double w = rsqrt(1.0 + s2);  // sqrt(rho'), rho' = 1/(1+s)
  B.rho0 = log(1.0 + s2);
  red0 = w * red0;
  red1 = w * red1;
  B.r[0] = w * r0;
  B.r[1] = w * r1;
  B.red[0] = red0;
  B.red[1] = red1;
  // u^T skew(v) = (u x v)^T
  d3 rm0 = vmul(red0, u.M2), rm1 = vmul(red1, u.M2);
  B.jti[0] = -cross(rm0, Xbi);  // reduce ric^T Rj^T Ri (-skew(pts_imu_i))         (:104-108)
  B.jti[1] = -cross(rm1, Xbi);
  d3 rr0 = vmul(red0, u.ricT), rr1 = vmul(red1, u.ricT);
  B.jtj[0] = cross(rr0, Xbj);  // reduce ric^T skew(pts_imu_j)                     (:116-120)
  B.jtj[1] = cross(rr1, Xbj);
  // reduce ric^T Rj^T Ri ric (:126,:137,:143): T is that product unless the pair is off the sphere — then (reduce M2) ric
  d3 rt0 = vmul(red0, u.T), rt1 = vmul(red1, u.T);
  if (OFFS && u.offc) rt0 = vmul(rm0, u.ric), rt1 = vmul(rm1, u.ric);
  // -T skew(Xci) + skew(T Xci) + skew(c)  ==  -T skew(Xci) + skew(Xcj)             (:126-131)
  B.jtx[0] = cross(red0, XcjJ) - cross(rt0, Xci);
  B.jtx[1] = cross(red1, XcjJ) - cross(rt1, Xci);
  const double il2 = inv_lam * inv_lam;
  B.jl[0] = -dot(rt0, pi) * il2;  // :137
  B.jl[1] = -dot(rt1, pi) * il2;
  if (est_td) {  // :143-144 as coded (velocity_j.head(2), not the true derivative)
    B.jtd[0] = -dot(rt0, ob.vi) * inv_lam + w * sqrt_info * ob.vj.x;
    B.jtd[1] = -dot(rt1, ob.vi) * inv_lam + w * sqrt_info * ob.vj.y;
  } else {
    B.jtd[0] = B.jtd[1] = 0.0;
  }
}

// Block-cooperative construction of the per-frame / per-pair table for one state.
// Per-frame and per-pair quantities of one linearization point.  poses: 12 x 7 doubles in LDS (pose[0..10], ex_pose);
// lds: >= TAB_SCRATCH doubles of scratch.  The intermediate results travel through LDS (a global write -> barrier -> read by
// another thread is a full memory round trip each time, and there are two of them), the table itself is written once.
// Needs >= 121 threads; every thread of the workgroup has to come here (barriers inside).
// A quaternion off the unit sphere (struct Tab, dev_types.h) gets its back-rotation R(q^-1) beside R^T; the pairs it touches
// take T and c from it and keep the transposed flavour of c for their Jacobians.
constexpr int TAB_SCRATCH = 376;
// (out of line: the few lanes that ever come here must not cost every kernel that builds a table their registers — k_setup of a
// resident batch runs five workgroups per CU)
__device__ __attribute__((noinline)) void off_sphere_back_rotation(double qw, double qx, double qy, double qz, double *RI, double *RF) {
  const m33 rI = q2R(qinv(q4{qw, qx, qy, qz}));
  stm(RI, rI);
  if (RF) stm(RF, inv33(rI));
}
DEV bool q_off_sphere(q4 q) { return fabs((q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z) - 1.0) > TAB_OFF_SPHERE; }
// DETECT = false (the candidates of a pass: every pose has just left PoseLocalParameterization::Plus, which normalizes; an extrinsic
// that is not estimated stays what it was): nothing is tested, the extrinsic is off the sphere iff ex_off says so (Slot::ex_fixed_off).
// Returns the table's off-sphere mask (every thread).
template <bool DETECT = true>
DEV unsigned build_tab(const double *poses, Tab *t, int tid, double *lds, int ex_off = 0) {
  double *R = lds, *P = lds + 99, *ric = lds + 132, *ricT = lds + 141, *tic = lds + 150, *M1 = lds + 153;  // .. 252
  double *RI = lds + 252, *ricI = lds + 351;  // R(q^-1) of the frames / the extrinsic whose quaternion is off the sphere
  unsigned *offm = (unsigned *)(lds + 360);   // their mask
  bool off = false;
  if (tid < 11) {
    const double *p = poses + 7 * tid;
    const q4 q = q_from_pose(p);
    const m33 r = q2R(q);
    stm(R + 9 * tid, r), stm(t->R[tid], r);
    off = DETECT && q_off_sphere(q);
    if (off) off_sphere_back_rotation(q.w, q.x, q.y, q.z, RI + 9 * tid, nullptr);
#pragma unroll
    for (int k = 0; k < 3; k++) P[3 * tid + k] = p[k], t->P[tid][k] = p[k];
  } else if (tid == 11) {
    const double *p = poses + 77;
    const q4 q = q_from_pose(p);
    const m33 r = q2R(q), rT = tr(r);
    stm(ric, r), stm(ricT, rT), stm(t->ric, r), stm(t->ricT, rT);
    off = DETECT ? q_off_sphere(q) : ex_off != 0;
    if (off) off_sphere_back_rotation(q.w, q.x, q.y, q.z, ricI, t->T[0]);  // ricF
    else stm(t->T[0], r);
#pragma unroll
    for (int k = 0; k < 3; k++) tic[k] = p[k], t->tic[k] = p[k];
  }
  if (DETECT) {
    if (tid < 64) {  // (the twelve quaternions sit in wave 0)
      const unsigned m = (unsigned)__ballot(off) & 0xfffu;
      if (tid == 0) *offm = m, t->c[0][0] = (double)m;
    }
  } else if (tid == 0) {
    const unsigned m = ex_off ? 1u << TAB_EX_BIT : 0u;
    *offm = m, t->c[0][0] = (double)m;
  }
  __syncthreads();
  if (tid < 11) {
    const m33 m = mm(ldm(ricT), tr(ldm(R + 9 * tid)));
    stm(M1 + 9 * tid, m), stm(t->M1[tid], m);
  }
  __syncthreads();
  if (tid < NPAIR) {
    const int i = tid / 11, j = tid % 11;
    if (i < j) {
      const m33 Ri = ldm(R + 9 * i), rc = ldm(ric);
      const m33 M2 = mm(ldm(M1 + 9 * j), Ri);
      stm(t->M2[tid], M2);
      const d3 tc = ld3(tic);
      const d3 v = mul(Ri, tc) + ld3(P + 3 * i) - ld3(P + 3 * j);
      const m33 RjT = tr(ldm(R + 9 * j));
      const d3 cc = mul(ldm(ricT), mul(RjT, v) - tc);
      const unsigned om = *offm;
      if (pair_is_off(om, j)) {
        // the residual chain of this pair through Quaternion::inverse() (projection_td_factor.cpp:59-60)
        const m33 RjI = (om >> j) & 1 ? ldm(RI + 9 * j) : RjT, rcI = (om >> TAB_EX_BIT) & 1 ? ldm(ricI) : ldm(ricT);
        stm(t->T[tid], mm(mm(rcI, RjI), mm(Ri, rc)));
        const d3 cr = mul(rcI, mul(RjI, v) - tc);
        t->c[tid][0] = cr.x, t->c[tid][1] = cr.y, t->c[tid][2] = cr.z;
        const int tj = tab_cj(tid);
        t->c[tj][0] = cc.x, t->c[tj][1] = cc.y, t->c[tj][2] = cc.z;
      } else {
        stm(t->T[tid], mm(M2, rc));
        t->c[tid][0] = cc.x, t->c[tid][1] = cc.y, t->c[tid][2] = cc.z;
      }
    }
  }
  __syncthreads();
  return *offm;
}

// ---------------------------------------------------------------------------
// IMU
// ---------------------------------------------------------------------------
DEV m33 jblk(const double *J15, int r0, int c0) {
  m33 b;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) b.a[i * 3 + j] = J15[(r0 + i) * 15 + c0 + j];
  return b;
}

// IntegrationBase::evaluate — raw (un-whitened) residual, integration_base.h:160-186
DEV void imu_raw_residual(const LfvioPreintegration *pre, const double *G, const double *pose_i, const double *sb_i,
                          const double *pose_j, const double *sb_j, double *r) {
  d3 g = ld3(G);
  d3 Pi = ld3(pose_i), Pj = ld3(pose_j);
  q4 Qi = q_from_pose(pose_i), Qj = q_from_pose(pose_j);
  d3 Vi = ld3(sb_i), Bai = ld3(sb_i + 3), Bgi = ld3(sb_i + 6);
  d3 Vj = ld3(sb_j), Baj = ld3(sb_j + 3), Bgj = ld3(sb_j + 6);
  m33 dp_dba = jblk(pre->jacobian, 0, 9), dp_dbg = jblk(pre->jacobian, 0, 12), dq_dbg = jblk(pre->jacobian, 3, 12);
  m33 dv_dba = jblk(pre->jacobian, 6, 9), dv_dbg = jblk(pre->jacobian, 6, 12);
  d3 dba = Bai - ld3(pre->linearized_ba), dbg = Bgi - ld3(pre->linearized_bg);
  q4 dq = q4{pre->delta_q[3], pre->delta_q[0], pre->delta_q[1], pre->delta_q[2]};
  const double dt = pre->sum_dt;
  q4 cq = qmul(dq, deltaQ(mul(dq_dbg, dbg)));
  d3 cv = ld3(pre->delta_v) + mul(dv_dba, dba) + mul(dv_dbg, dbg);
  d3 cp = ld3(pre->delta_p) + mul(dp_dba, dba) + mul(dp_dbg, dbg);
  q4 Qi_inv = qinv(Qi);
  d3 rp = qrot(Qi_inv, (0.5 * dt * dt) * g + Pj - Pi - dt * Vi) - cp;
  q4 qr = qmul(qinv(cq), qmul(Qi_inv, Qj));
  d3 rv = qrot(Qi_inv, dt * g + Vj - Vi) - cv;
  r[0] = rp.x, r[1] = rp.y, r[2] = rp.z;
  r[3] = 2.0 * qr.x, r[4] = 2.0 * qr.y, r[5] = 2.0 * qr.z;
  r[6] = rv.x, r[7] = rv.y, r[8] = rv.z;
  r[9] = Baj.x - Bai.x, r[10] = Baj.y - Bai.y, r[11] = Baj.z - Bai.z;
  r[12] = Bgj.x - Bgi.x, r[13] = Bgj.y - Bgi.y, r[14] = Bgj.z - Bgi.z;
}

DEV void put33(double *J, int ld, int r0, int c0, const m33 &b, double sgn) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) J[(r0 + i) * ld + c0 + j] = sgn * b.a[i * 3 + j];
}
// bottom-right 3x3 of Qleft(q) = w I + skew(v);  of Qleft(a) Qright(b):
//   -(va vb^T) + (wa I + skew(va)) (wb I - skew(vb))
DEV m33 qleft_br(q4 q) {
  m33 s = skewm(qvec(q));
  s.a[0] += q.w, s.a[4] += q.w, s.a[8] += q.w;
  return s;
}
DEV m33 qleft_qright_br(q4 a, q4 b) {
  m33 L = qleft_br(a);
  m33 R = skewm(qvec(b));
#pragma unroll
  for (int i = 0; i < 9; i++) R.a[i] = -R.a[i];
  R.a[0] += b.w, R.a[4] += b.w, R.a[8] += b.w;
  m33 P = mm(L, R);
  d3 va = qvec(a), vb = qvec(b);
  const double av[3] = {va.x, va.y, va.z}, bv[3] = {vb.x, vb.y, vb.z};
  // row 1..3 of Qleft(a) col 0 is va; row 0 of Qright(b) cols 1..3 is -vb^T
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) P.a[i * 3 + j] -= av[i] * bv[j];
  return P;
}

// raw Jacobian 15 x 30 in local coordinates [pose_i(6) sb_i(9) pose_j(6) sb_j(9)]; Jraw zeroed by caller.
// imu_factor.h:88-196 (before the sqrt_info multiplication)
DEV void imu_raw_jacobian(const LfvioPreintegration *pre, const double *G, const double *pose_i, const double *sb_i,
                          const double *pose_j, const double *sb_j, double *J, const int LD = 30) {
  d3 g = ld3(G);
  d3 Pi = ld3(pose_i), Pj = ld3(pose_j);
  q4 Qi = q_from_pose(pose_i), Qj = q_from_pose(pose_j);
  d3 Vi = ld3(sb_i), Bgi = ld3(sb_i + 6);
  d3 Vj = ld3(sb_j);
  const double dt = pre->sum_dt;
  m33 dp_dba = jblk(pre->jacobian, 0, 9), dp_dbg = jblk(pre->jacobian, 0, 12), dq_dbg = jblk(pre->jacobian, 3, 12);
  m33 dv_dba = jblk(pre->jacobian, 6, 9), dv_dbg = jblk(pre->jacobian, 6, 12);
  q4 dq = q4{pre->delta_q[3], pre->delta_q[0], pre->delta_q[1], pre->delta_q[2]};
  q4 cq = qmul(dq, deltaQ(mul(dq_dbg, Bgi - ld3(pre->linearized_bg))));
  q4 Qi_inv = qinv(Qi);
  m33 RiT = q2R(Qi_inv);
  m33 I = skewm(mk3(0, 0, 0));
  I.a[0] = I.a[4] = I.a[8] = 1.0;
  // pose_i: cols 0..5
  put33(J, LD, 0, 0, RiT, -1.0);
  put33(J, LD, 0, 3, skewm(qrot(Qi_inv, (0.5 * dt * dt) * g + Pj - Pi - dt * Vi)), 1.0);
  put33(J, LD, 3, 3, qleft_qright_br(qmul(qinv(Qj), Qi), cq), -1.0);
  put33(J, LD, 6, 3, skewm(qrot(Qi_inv, dt * g + Vj - Vi)), 1.0);
  // sb_i: cols 6..14
  put33(J, LD, 0, 6, RiT, -dt);
  put33(J, LD, 0, 9, dp_dba, -1.0);
  put33(J, LD, 0, 12, dp_dbg, -1.0);
  put33(J, LD, 3, 12, mm(qleft_br(qmul(qmul(qinv(Qj), Qi), dq)), dq_dbg), -1.0);
  put33(J, LD, 6, 6, RiT, -1.0);
  put33(J, LD, 6, 9, dv_dba, -1.0);
  put33(J, LD, 6, 12, dv_dbg, -1.0);
  put33(J, LD, 9, 9, I, -1.0);
  put33(J, LD, 12, 12, I, -1.0);
  // pose_j: cols 15..20
  put33(J, LD, 0, 15, RiT, 1.0);
  put33(J, LD, 3, 18, qleft_br(qmul(qmul(qinv(cq), Qi_inv), Qj)), 1.0);
  // sb_j: cols 21..29
  put33(J, LD, 6, 21, RiT, 1.0);
  put33(J, LD, 9, 24, I, 1.0);
  put33(J, LD, 12, 27, I, 1.0);
}

// MarginalizationFactor::Evaluate dx part (marginalization_factor.cpp:343-362) for one prior block: kind, frame, first
// column idx and linearization point x0 as uploaded (Slot::prior_kind / prior_frame / prior_idx / prior_x0).
DEV void prior_block_dx(int kind, int frame, int idx, const double *x0, const FrameState *x, double *dx) {
  const double *xb = kind == LFVIO_BLOCK_POSE ? x->pose[frame]
                     : kind == LFVIO_BLOCK_SPEEDBIAS ? x->sb[frame]
                     : kind == LFVIO_BLOCK_EX_POSE ? x->ex
                                                   : &x->td;
  if (kind == LFVIO_BLOCK_POSE || kind == LFVIO_BLOCK_EX_POSE) {
    dx[idx + 0] = xb[0] - x0[0], dx[idx + 1] = xb[1] - x0[1], dx[idx + 2] = xb[2] - x0[2];
    q4 dq = qmul(qinv(q4{x0[6], x0[3], x0[4], x0[5]}), q_from_pose(xb));
    double s = (dq.w >= 0) ? 2.0 : -2.0;
    dx[idx + 3] = s * dq.x, dx[idx + 4] = s * dq.y, dx[idx + 5] = s * dq.z;
  } else {
    const int sz = kind == LFVIO_BLOCK_SPEEDBIAS ? 9 : 1;
#pragma unroll
    for (int k = 0; k < 9; k++)  // (static indices: x0 may live in registers)
      if (k < sz) dx[idx + k] = xb[k] - x0[k];
  }
}
DEV void prior_block_dx(const Slot *S, const FrameState *x, int bi, double *dx) {
  prior_block_dx(S->prior_kind[bi], S->prior_frame[bi], S->prior_idx[bi], S->prior_x0[bi], x, dx);
}
