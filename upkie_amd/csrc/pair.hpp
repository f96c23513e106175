// pair.hpp -- two lanes per env: each lane owns one leg (left = even lane,
// right = odd lane) and a redundant copy of the base.
//
// Small batches are latency bound: at 4096 envs the one-env-per-lane kernel
// puts 64 wavefronts on a chip with 1024 SIMDs and the launch lasts as long as
// ONE wave's instruction stream. Almost half of that stream is per-leg work
// (kinematics, Newton-Euler, composite inertias, the 3x3 leg block, the tire's
// contact rows), which two lanes can do side by side. Everything that couples
// the legs goes through the 6x6 base system, so the lanes only meet in a few
// sums and in the contact matrix: ~100 DPP swaps (quad_perm [1,0,3,2], one VALU
// instruction each, no LDS) per substep. Sums are formed as own + partner,
// which is commutative in IEEE arithmetic, and the one non-symmetric product
// (the left/right block of the contact matrix) is computed by the right lane
// and copied, so both lanes hold bit-identical base quantities throughout.
//
// Included by upkie_hip.hip inside namespace upkie, after the one-lane kernel
// (it reuses DevConfig, Servo, joint_torque, sample_init_state, philox_*).
#pragma once

// value held by the partner lane (lane ^ 1)
__device__ __forceinline__ float xchg(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
// Where the lane that owns the other leg sits (a policy, so that the substep below does not depend on the lane layout).
struct AdjacentLanes {
  static __device__ __forceinline__ float xchg(float x) { return upkie::xchg(x); }
};
template <class X = AdjacentLanes>
__device__ __forceinline__ float pair_sum(float x) { return x + X::xchg(x); }
template <class X = AdjacentLanes>
__device__ __forceinline__ V3 pair_sum(V3 a) { return V3{pair_sum<X>(a.x), pair_sum<X>(a.y), pair_sum<X>(a.z)}; }
// (the exchange comes first: under `b || xchg(...)` a lane whose b is true would skip the DPP
// move and its partner would read a disabled lane)
template <class X = AdjacentLanes>
__device__ __forceinline__ bool pair_any(bool b) {
  const float partner = X::xchg(b ? 1.f : 0.f);
  return b || partner != 0.f;
}
template <class T>
__device__ __forceinline__ T pick(int leg, T left, T right) {
  return leg ? right : left;
}

// Per-lane constants of the leg a lane owns.
struct PairLeg {
  LegRegs regs;
  float damping[3], lower[3], upper[3], effort[3], velocity[3], friction[3], control_noise[3], measurement_noise[3];
  int bounded[3];
  float wheel_center[3];
};

template <class ModelT>
__device__ __forceinline__ PairLeg load_pair_leg(const ModelT& M, const DevLimits& Lm, const DevConfig& C, int leg, const float* records,
                                                 size_t stride) {
  // the lane's row of DevModel::leg_table, 16 x 16 bytes (cached: two rows for the whole grid)
  float t[LT_WORDS];
  {
    typedef float Vec4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) Vec4* GlobalVec;
    GlobalVec row = (GlobalVec)(const void*)&M.leg_table[leg][0];
#pragma unroll
    for (int i = 0; i < LT_WORDS / 4; ++i) {
      const Vec4 v = row[i];
      t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w;
    }
  }
  PairLeg P;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int jl = k, jr = 3 + k;
    P.regs.m[k] = t[LT_MASS + k];
    P.regs.sg[k] = t[LT_SIGN + k];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      P.regs.c[k][d] = t[LT_COM + 3 * k + d];
      P.regs.p[k][d] = t[LT_POS + 3 * k + d];
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) P.regs.I[k][d] = t[LT_INERTIA + 6 * k + d];
    P.damping[k] = t[LT_DAMPING + k];
    P.lower[k] = t[LT_LOWER + k];
    P.upper[k] = t[LT_UPPER + k];
    P.bounded[k] = t[LT_BOUNDED + k] != 0.f;
    P.effort[k] = t[LT_EFFORT + k];
    P.velocity[k] = t[LT_VELOCITY + k];
    P.friction[k] = pick(leg, C.joint_friction[jl], C.joint_friction[jr]);
    P.control_noise[k] = pick(leg, C.control_noise[jl], C.control_noise[jr]);
    P.measurement_noise[k] = pick(leg, C.measurement_noise[jl], C.measurement_noise[jr]);
    P.wheel_center[k] = t[LT_WHEEL_CENTER + k];
  }
  if (records) {  // this env's inertial records of the owned leg's bodies (randomize_inertias)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float* r = records + (size_t)(UPKIE_INERTIAL_WORDS * (1 + 3 * leg + k)) * stride;
      P.regs.m[k] = r[0];
#pragma unroll
      for (int d = 0; d < 3; ++d) P.regs.c[k][d] = r[(size_t)(1 + d) * stride];
#pragma unroll
      for (int d = 0; d < 6; ++d) P.regs.I[k][d] = r[(size_t)(4 + d) * stride];
    }
  }
  (void)Lm;
  return P;
}

// The trunk's inertial record of one env (both lanes hold it).
struct TrunkInertial {
  float m;
  V3 c;
  S3 I;
};

struct PhysPair {
  V3 pos;
  float qw, qx, qy, qz;
  V3 linvel, angvel;
  float q[3], qd[3];  // the owned leg
};

// One physics substep, two lanes per env. Mirrors physics_substep() step by
// step; comments there apply. `leg` = 0 (left) / 1 (right) is the leg this lane
// owns. Returns the floor-contact flag (identical in both lanes).
template <bool ACTIVE_SET = false, class XL = AdjacentLanes, class ModelT>
__device__ __forceinline__ bool physics_substep_pair(const ModelT& M, const DevLimits& Lm, const PairLeg& PL, int leg, PhysPair& s,
                                                     const float (&tau)[3], float h, const TrunkInertial* trunk, const ExtForces& ext,
                                                     SweepWarmStart* warm = nullptr) {
  bool own_limit = false;
  if (Lm.enforce) {
#pragma unroll
    for (int k = 0; k < 3; ++k) own_limit = own_limit || joint_limit_near(PL.bounded[k] != 0, s.q[k], PL.lower[k], PL.upper[k], joint_limit_reach(s.qd[k], M.max_joint_velocity, h));
  }
  const bool any_limit = Lm.enforce ? pair_any<XL>(own_limit) : false;

  const BaseFrame bf = base_frame(s.qw, s.qx, s.qy, s.qz, s.linvel, s.angvel);
  const float r00 = bf.r00, r01 = bf.r01, r02 = bf.r02, r10 = bf.r10, r11 = bf.r11, r12 = bf.r12, r20 = bf.r20, r21 = bf.r21, r22 = bf.r22;
  const V3 vB = bf.vB, wB = bf.wB, nB = bf.nB;
  V3 gn = M.gravity * nB;

  // trunk (both lanes, identical)
  float m0 = trunk ? trunk->m : M.mass[0];
  V3 c0 = trunk ? trunk->c : v3(M.com[0][0], M.com[0][1], M.com[0][2]);
  S3 I0 = trunk ? trunk->I : S3{M.inertia[0][0], M.inertia[0][1], M.inertia[0][2], M.inertia[0][3], M.inertia[0][4], M.inertia[0][5]};
  V3 I0w = mul(I0, wB);
  V3 bias_f, bias_n;
  {
    V3 ac = cross(wB, cross(wB, c0));
    V3 f = m0 * (ac + gn);
    V3 n = cross(wB, I0w);
    bias_f = f;
    bias_n = n + cross(c0, f);
  }
  Composite total{m0, m0 * c0, shift_to_origin(I0, m0, c0)};

  // own leg, then legs' totals as own + partner
  Leg G;
  Composite legs{0.f, v3(0.f, 0.f, 0.f), S3{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
  V3 lf = v3(0.f, 0.f, 0.f), ln = v3(0.f, 0.f, 0.f);
  leg_pass(PL.regs, M.wheel_axisymmetric != 0, s.q, s.qd, wB, gn, G, legs, lf, ln);
  total.m += pair_sum<XL>(legs.m);
  total.h = total.h + pair_sum<XL>(legs.h);
  total.I = total.I + S3{pair_sum<XL>(legs.I.xx), pair_sum<XL>(legs.I.yy), pair_sum<XL>(legs.I.zz),
                         pair_sum<XL>(legs.I.xy), pair_sum<XL>(legs.I.xz), pair_sum<XL>(legs.I.yz)};
  bias_f = bias_f + pair_sum<XL>(lf);
  bias_n = bias_n + pair_sum<XL>(ln);

  Ldl6 fac;
  {
    float A[21];
    V3 hh = total.h;
    A[0] = total.m;
    A[1] = 0.f; A[2] = total.m;
    A[3] = 0.f; A[4] = 0.f; A[5] = total.m;
    A[6] = 0.f;   A[7] = -hh.z; A[8] = hh.y;  A[9] = total.I.xx;
    A[10] = hh.z; A[11] = 0.f;  A[12] = -hh.x; A[13] = total.I.xy; A[14] = total.I.yy;
    A[15] = -hh.y; A[16] = hh.x; A[17] = 0.f;  A[18] = total.I.xz; A[19] = total.I.yz; A[20] = total.I.zz;
    int idx = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        const float own = G.D[r][0] * G.F[0][c] + G.D[r][1] * G.F[1][c] + G.D[r][2] * G.F[2][c];
        A[idx] -= pair_sum<XL>(own);
        ++idx;
      }
    }
    ldl6_factor(A, fac);
  }

  // impulses so far: t = h (applied - bias)
  float tb[6], tl[3], rt[6];
  {
    V3 vc = vB + cross(wB, c0);
    float vn = fast_sqrt(dot(vc, vc)), wn = fast_sqrt(dot(wB, wB));
    float kl = M.base_linear_damping, ka = M.base_angular_damping;
    V3 F = (-m0 * (kl + kl * vn)) * vc;
    V3 T = (-(ka + ka * wn)) * I0w;
    V3 Ntot = T + cross(c0, F);
    float te[3] = {0.f, 0.f, 0.f};
    if (ext.force) {
      for (int i = 0; i < ext.slots->count; ++i) {
        const V3 f = v3(ext.force[(size_t)(3 * i) * ext.stride], ext.force[(size_t)(3 * i + 1) * ext.stride],
                        ext.force[(size_t)(3 * i + 2) * ext.stride]);
        const int b = ext.slots->body[i];
        const bool local = ext.slots->local[i] != 0;
        const V3 pt = v3(ext.slots->point[i][0], ext.slots->point[i][1], ext.slots->point[i][2]);
        V3 Fe = local ? f : v3(r00 * f.x + r10 * f.y + r20 * f.z, r01 * f.x + r11 * f.y + r21 * f.z, r02 * f.x + r12 * f.y + r22 * f.z);
        if (b == 0) {  // trunk: identical in both lanes
          F = F + Fe;
          Ntot = Ntot + cross(pt, Fe);
        } else {  // a leg link: the lane owning the leg computes, both add own + partner
          const bool mine = (b >= 4) == (leg == 1);
          V3 pe = pt;
          float t3[3] = {0.f, 0.f, 0.f};
          ext_on_leg(G, s.q, b >= 4 ? b - 4 : b - 1, local, pt, Fe, pe, t3);
          V3 Ne = cross(pe, Fe);
          if (!mine) {
            Fe = v3(0.f, 0.f, 0.f);
            Ne = v3(0.f, 0.f, 0.f);
          }
          F = F + pair_sum<XL>(Fe);
          Ntot = Ntot + pair_sum<XL>(Ne);
#pragma unroll
          for (int j = 0; j < 3; ++j) te[j] += mine ? t3[j] : 0.f;
        }
      }
    }
    tb[0] = h * (F.x - bias_f.x); tb[1] = h * (F.y - bias_f.y); tb[2] = h * (F.z - bias_f.z);
    tb[3] = h * (Ntot.x - bias_n.x); tb[4] = h * (Ntot.y - bias_n.y); tb[5] = h * (Ntot.z - bias_n.z);
#pragma unroll
    for (int k = 0; k < 3; ++k) tl[k] = h * (tau[k] + te[k] - PL.damping[k] * s.qd[k] - G.bias[k]);
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) rt[c] = tb[c] - pair_sum<XL>(G.D[c][0] * tl[0] + G.D[c][1] * tl[1] + G.D[c][2] * tl[2]);

  // ---- contact rows of the owned tire ------------------------------------
  float Jb[3][6], Jl[3][3], Jt[3][6], vnow[3];
  float un = fast_sqrt(nB.x * nB.x + nB.z * nB.z);
  float iun = fast_rcp(fmaxf(un, 1e-12f));
  float denom = h * M.contact_stiffness + M.contact_damping;
  float ih = fast_rcp(h);
  float erp = denom > 0.f ? h * M.contact_stiffness * fast_rcp(denom) : 0.2f;
  float cfm = denom > 0.f ? fast_rcp(denom * h) : 0.f;
  float dist;
  bool active_own;
  {
    float sa = G.sgn[2];
    V3 center = G.o[2] + v3(PL.wheel_center[0], PL.wheel_center[1], PL.wheel_center[2]);
    V3 dlow = v3(-nB.x * iun, 0.f, -nB.z * iun);
    V3 P = center + M.wheel_radius * dlow;
    dist = s.pos.z + dot(nB, P);
    active_own = un >= 1e-6f && dist <= M.contact_breaking_threshold;
    V3 t1 = (sa * iun) * v3(nB.z, 0.f, -nB.x);
    V3 t2 = cross(nB, t1);
    V3 dirs[3] = {nB, t1, t2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      V3 d = dirs[k];
      V3 Pxd = cross(P, d);
      Jb[k][0] = d.x; Jb[k][1] = d.y; Jb[k][2] = d.z;
      Jb[k][3] = Pxd.x; Jb[k][4] = Pxd.y; Jb[k][5] = Pxd.z;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        V3 rr = P - G.o[j];
        Jl[k][j] = G.sgn[j] * (rr.z * d.x - rr.x * d.z);
      }
      float v = Jb[k][0] * vB.x + Jb[k][1] * vB.y + Jb[k][2] * vB.z + Jb[k][3] * wB.x + Jb[k][4] * wB.y + Jb[k][5] * wB.z;
#pragma unroll
      for (int j = 0; j < 3; ++j) v = fmaf(Jl[k][j], s.qd[j], v);
      vnow[k] = v;
#pragma unroll
      for (int c = 0; c < 6; ++c) Jt[k][c] = Jb[k][c] - (G.D[c][0] * Jl[k][0] + G.D[c][1] * Jl[k][1] + G.D[c][2] * Jl[k][2]);
    }
  }
  const bool active_partner = XL::xchg(active_own ? 1.f : 0.f) != 0.f;
  const bool active_l = pick(leg, active_own, active_partner), active_r = pick(leg, active_partner, active_own);
  const bool any_contact = active_own || active_partner;

  if (warm && (any_limit || !any_contact)) warm->swept = 0;
  if (any_limit) {
    // rare: rebuild both legs' data in both lanes and run the shared path
    System S2;
    S2.A = fac;
    Leg Gp = G;  // only Hinv and D of the partner are needed
#pragma unroll
    for (int i = 0; i < 6; ++i) Gp.Hinv[i] = XL::xchg(G.Hinv[i]);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Gp.D[r][c] = XL::xchg(G.D[r][c]);
    S2.leg[0] = pick(leg, G, Gp);
    S2.leg[1] = pick(leg, Gp, G);
    float lo6[6], up6[6], q6[6], qd6[6], vn6[6], Jt6[6][6], Jb6[6][6], Jl6[6][3], d2[2], tl2[3], tr2[3];
    int bd6[6];
    bool act2[2] = {active_l, active_r};
    const float pdist = XL::xchg(dist);
    d2[0] = pick(leg, dist, pdist);
    d2[1] = pick(leg, pdist, dist);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float plo = XL::xchg(PL.lower[k]), pup = XL::xchg(PL.upper[k]), pq = XL::xchg(s.q[k]), pqd = XL::xchg(s.qd[k]), pvn = XL::xchg(vnow[k]);
      const float pb = XL::xchg(PL.bounded[k] ? 1.f : 0.f), ptl = XL::xchg(tl[k]);
      lo6[k] = pick(leg, PL.lower[k], plo); lo6[3 + k] = pick(leg, plo, PL.lower[k]);
      up6[k] = pick(leg, PL.upper[k], pup); up6[3 + k] = pick(leg, pup, PL.upper[k]);
      bd6[k] = pick(leg, PL.bounded[k], (int)(pb != 0.f)); bd6[3 + k] = pick(leg, (int)(pb != 0.f), PL.bounded[k]);
      q6[k] = pick(leg, s.q[k], pq); q6[3 + k] = pick(leg, pq, s.q[k]);
      qd6[k] = pick(leg, s.qd[k], pqd); qd6[3 + k] = pick(leg, pqd, s.qd[k]);
      vn6[k] = pick(leg, vnow[k], pvn); vn6[3 + k] = pick(leg, pvn, vnow[k]);
      tl2[k] = pick(leg, tl[k], ptl); tr2[k] = pick(leg, ptl, tl[k]);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const float pjt = XL::xchg(Jt[k][c]), pjb = XL::xchg(Jb[k][c]);
        Jt6[k][c] = pick(leg, Jt[k][c], pjt); Jt6[3 + k][c] = pick(leg, pjt, Jt[k][c]);
        Jb6[k][c] = pick(leg, Jb[k][c], pjb); Jb6[3 + k][c] = pick(leg, pjb, Jb[k][c]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float pjl = XL::xchg(Jl[k][j]);
        Jl6[k][j] = pick(leg, Jl[k][j], pjl); Jl6[3 + k][j] = pick(leg, pjl, Jl[k][j]);
      }
    }
    float contact_lam[6];
    limit_path(M, S2, lo6, up6, bd6, q6, qd6, Jt6, Jb6, Jl6, vn6, d2, act2, cfm, erp, ih, M.max_joint_velocity, h, rt, tb, tl2, tr2, contact_lam);
#pragma unroll
    for (int k = 0; k < 3; ++k) tl[k] = pick(leg, tl2[k], tr2[k]);
  } else if (any_contact) {
    // own rows: Y = A^-1 Jt, K = Hinv J_leg, free velocity, own diagonal block
    float Y[3][6], K[3][3], rhs_own[3], Dg[6];
    const float* hv = G.Hinv;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
      for (int c = 0; c < 6; ++c) Y[b][c] = Jt[b][c];
      ldl6_solve(fac, Y[b]);
      K[b][0] = hv[0] * Jl[b][0] + hv[3] * Jl[b][1] + hv[4] * Jl[b][2];
      K[b][1] = hv[3] * Jl[b][0] + hv[1] * Jl[b][1] + hv[5] * Jl[b][2];
      K[b][2] = hv[4] * Jl[b][0] + hv[5] * Jl[b][1] + hv[2] * Jl[b][2];
      float vf = vnow[b];
#pragma unroll
      for (int c = 0; c < 6; ++c) vf = fmaf(Y[b][c], rt[c], vf);
      vf += K[b][0] * tl[0] + K[b][1] * tl[1] + K[b][2] * tl[2];
      const float rb = (b == 0) ? (dist <= 0.f ? -vf + erp * (-dist) * ih : -vf - dist * ih) : -vf;
      rhs_own[b] = active_own ? rb : 0.f;
#pragma unroll
      for (int a = b; a < 3; ++a) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc = fmaf(Jt[a][c], Y[b][c], acc);
        acc += Jl[a][0] * K[b][0] + Jl[a][1] * K[b][1] + Jl[a][2] * K[b][2];
        if (a == b) acc += a == 0 ? cfm : M.friction_cfm;
        Dg[a * (a + 1) / 2 + b] = acc;
      }
    }
    // cross block: rows of the RIGHT tire against columns of the LEFT tire,
    // X[a][b] = Jt_R[a] . Y_L[b]; computed by the right lane, copied to the left
    float X[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float pY[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) pY[c] = XL::xchg(Y[b][c]);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc = fmaf(Jt[a][c], pY[c], acc);
        X[a][b] = acc;
      }
    }
    float A[21], rhs[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float px = XL::xchg(X[a][b]);
        const float v = pick(leg, px, X[a][b]);  // the right lane's product
        A[(3 + a) * (4 + a) / 2 + b] = (active_l && active_r) ? v : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float pd = XL::xchg(Dg[i]);
      // packed index i of a 3x3 lower triangle: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
      const int a = i < 1 ? 0 : (i < 3 ? 1 : 2), b = i - a * (a + 1) / 2;
      const float dl = pick(leg, Dg[i], pd), dr = pick(leg, pd, Dg[i]);
      A[a * (a + 1) / 2 + b] = active_l ? dl : (a == b ? 1.f : 0.f);
      A[(3 + a) * (4 + a) / 2 + 3 + b] = active_r ? dr : (a == b ? 1.f : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float pr = XL::xchg(rhs_own[k]);
      rhs[k] = pick(leg, rhs_own[k], pr);
      rhs[3 + k] = pick(leg, pr, rhs_own[k]);
    }
    float lam[6];
    const float mu = M.friction_mu;
    bool need_pgs = false;
    {
      Ldl6 cf;
      ldl6_factor(A, cf);
#pragma unroll
      for (int r = 0; r < 6; ++r) lam[r] = rhs[r];
      ldl6_solve(cf, lam);
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        if (lam[3 * w] < 0.f) {
          lam[3 * w] = 0.f;
          need_pgs = true;
        }
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        if ((r % 3) == 0) continue;
        const float lim = mu * lam[3 * (r / 3)];
        if (lam[r] < -lim) { lam[r] = -lim; need_pgs = true; }
        if (lam[r] > lim) { lam[r] = lim; need_pgs = true; }
      }
    }
    if (need_pgs) {
      contact_sweeps_warm<ACTIVE_SET>(M, A, rhs, lam, active_l && active_r, warm);  // identical data in both lanes: they iterate in lockstep
    } else if (warm) {
      warm->swept = 0;
    }
    // t += J' lam: own rows locally, base part as own + partner
    float lo_[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) lo_[k] = pick(leg, lam[k], lam[3 + k]);
#pragma unroll
    for (int c = 0; c < 6; ++c) tb[c] += pair_sum<XL>(Jb[0][c] * lo_[0] + Jb[1][c] * lo_[1] + Jb[2][c] * lo_[2]);
#pragma unroll
    for (int j = 0; j < 3; ++j) tl[j] += Jl[0][j] * lo_[0] + Jl[1][j] * lo_[1] + Jl[2][j] * lo_[2];
  }

  // nu+ = nu + M^-1 t
  float xb[6], xl[3];
#pragma unroll
  for (int c = 0; c < 6; ++c) xb[c] = tb[c] - pair_sum<XL>(G.D[c][0] * tl[0] + G.D[c][1] * tl[1] + G.D[c][2] * tl[2]);
  ldl6_solve(fac, xb);
  {
    const float* hv = G.Hinv;
    xl[0] = hv[0] * tl[0] + hv[3] * tl[1] + hv[4] * tl[2];
    xl[1] = hv[3] * tl[0] + hv[1] * tl[1] + hv[5] * tl[2];
    xl[2] = hv[4] * tl[0] + hv[5] * tl[1] + hv[2] * tl[2];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      xl[0] -= G.D[r][0] * xb[r];
      xl[1] -= G.D[r][1] * xb[r];
      xl[2] -= G.D[r][2] * xb[r];
    }
  }
  const float n0 = vB.x + xb[0], n1 = vB.y + xb[1], n2 = vB.z + xb[2];
  const float n3 = wB.x + xb[3], n4 = wB.y + xb[4], n5 = wB.z + xb[5];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float v = fminf(fmaxf(s.qd[j] + xl[j], -M.max_joint_velocity), M.max_joint_velocity);
    s.qd[j] = v;
    s.q[j] = fmaf(h, v, s.q[j]);
  }
  integrate_base(bf, n0, n1, n2, n3, n4, n5, h, s.pos, s.qw, s.qx, s.qy, s.qz, s.linvel, s.angvel);
  return any_contact;
}

// Which two-lane instantiations address the state through its buffer descriptor (state_words.hpp): the multi-step ones
// (their step is a loop body: lane addresses hoisted across it are what they spill); an experiment may force it either way
#if defined(UPKIE_PAIR_BUFFERED_STATE)
constexpr bool pair_state_through_descriptor(int) { return UPKIE_PAIR_BUFFERED_STATE != 0; }
#else
constexpr bool pair_state_through_descriptor(int mode) { return mode == MODE_PENDULUM_ROLLOUT; }
#endif

// One env.step() of B envs on 2 B lanes. Same contract as step_kernel.
template <int MODE, bool RAND, bool SPINE>
__global__ __launch_bounds__(64) void step_kernel_pair(const DevModel* __restrict__ Mp, DevLimits Lm, DevConfig C,
                                                        float* __restrict__ state, const float* __restrict__ act,
                                                        float* __restrict__ obs, float* __restrict__ reward,
                                                        uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                                                        const uint8_t* __restrict__ mask, const float* __restrict__ body_inertials,
                                                        const float* __restrict__ ext_force, int packed, BaseVelocityPtrs bv,
                                                        float* __restrict__ spine_state, float* __restrict__ final_obs, int n_steps) {
  warm_kernel_arguments();
  typedef const __attribute__((address_space(4))) DevModel* ConstModelPtr;
  const int B = C.num_envs;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = tid >> 1, leg = tid & 1;
  // UpkieBaseVelocity with its MPC balancer in the same launch (upkie_sim_step_base_velocity_mpc): the wavefront first
  // solves the condensed QPs of its 32 envs, 16 at a time on the matrix cores (mpc_tile, all 64 lanes), and hands the
  // commanded velocities to the lanes that step those envs through LDS
  __shared__ float mpc_velocity[MODE == MODE_BASE_VELOCITY ? 32 : 1];
  if (MODE == MODE_BASE_VELOCITY && bv.mpc_fused) {
    const int env0 = 32 * blockIdx.x;
    const float* done_row = C.autoreset_mode != 0 ? state + (size_t)UPKIE_S_DONE * B : nullptr;
#if defined(UPKIE_FUSED_MPC_FP32)
    mpc_tile<1>(bv.mpc, bv.mpc_ws, bv.x0, act, 2, bv.contact, done_row, C.dt, bv.mpc_commanded, nullptr, env0, mpc_velocity);
    mpc_tile<1>(bv.mpc, bv.mpc_ws, bv.x0, act, 2, bv.contact, done_row, C.dt, bv.mpc_commanded, nullptr, env0 + 16, mpc_velocity + 16);
#else
    mpc_tile_h<1>(bv.mpc, bv.mpc_ws, bv.x0, act, 2, bv.contact, done_row, C.dt, bv.mpc_commanded, nullptr, env0, mpc_velocity);
    mpc_tile_h<1>(bv.mpc, bv.mpc_ws, bv.x0, act, 2, bv.contact, done_row, C.dt, bv.mpc_commanded, nullptr, env0 + 16, mpc_velocity + 16);
#endif
    __syncthreads();
  }
  if (e >= B) return;  // both lanes of a pair leave together
  const bool lead = leg == 0;  // the lane that writes per-env (not per-leg) words
  // state word w of env e: through 64-bit lane addresses or -- BUFFERED -- the state's buffer descriptor (state_words.hpp);
  // the per-leg blocks (three joints, two low-pass targets) start at the own leg's row
  constexpr bool BUFFERED = pair_state_through_descriptor(MODE);
  const unsigned row_bytes = (unsigned)B * 4u;
  const __amdgpu_buffer_rsrc_t state_rsrc = state_descriptor(state, B);
  float* const st = state + e;
  const unsigned env_off = (unsigned)e * 4u;
  const unsigned leg3_off = env_off + (unsigned)(3 * leg) * row_bytes, leg2_off = env_off + (unsigned)(2 * leg) * row_bytes;
#define SWI(w, i, off) state_word<BUFFERED>(state_rsrc, st, (size_t)((w) + (i)) * B, (off), (unsigned)(w) * row_bytes)
#define SW(w) SWI(w, 0, env_off)

  // ---- load ----------------------------------------------------------
  PhysPair s;
  s.pos = v3(SW(UPKIE_S_POS), SW(UPKIE_S_POS + 1), SW(UPKIE_S_POS + 2));
  s.qw = SW(UPKIE_S_QUAT); s.qx = SW(UPKIE_S_QUAT + 1); s.qy = SW(UPKIE_S_QUAT + 2); s.qz = SW(UPKIE_S_QUAT + 3);
  s.linvel = v3(SW(UPKIE_S_LINVEL), SW(UPKIE_S_LINVEL + 1), SW(UPKIE_S_LINVEL + 2));
  s.angvel = v3(SW(UPKIE_S_ANGVEL), SW(UPKIE_S_ANGVEL + 1), SW(UPKIE_S_ANGVEL + 2));
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    s.q[k] = SWI(UPKIE_S_Q + k, 3 * leg, leg3_off);
    s.qd[k] = SWI(UPKIE_S_QD + k, 3 * leg, leg3_off);
  }
  float legref[2] = {SWI(UPKIE_S_LEGREF, 2 * leg, leg2_off), SWI(UPKIE_S_LEGREF + 1, 2 * leg, leg2_off)};
  constexpr bool YAWING = MODE == MODE_GYROPOD || MODE == MODE_BASE_VELOCITY;
  float yaw = 0.f, yawvel = 0.f;
  if (YAWING) {
    yaw = SW(UPKIE_S_YAW);
    yawvel = SW(UPKIE_S_YAWVEL);
  }
  const ExtForces ext{RAND && ext_force ? ext_force + e : nullptr, (size_t)B, &C.ext};
  // the owned leg's constants: selected once, kept in registers for the launch
  const float* records = RAND && body_inertials ? body_inertials + e : nullptr;
  const PairLeg PL = load_pair_leg(*(ConstModelPtr)Mp, Lm, C, leg, records, (size_t)B);
  // uniform model constants through the constant address space: scalar loads issued with the
  // kernel arguments, not vector loads that wait behind the state
  const auto& M = *(ConstModelPtr)Mp;
  TrunkInertial trunk;
  if (RAND) {
    if (records) {
      trunk.m = records[0];
      trunk.c = v3(records[(size_t)1 * B], records[(size_t)2 * B], records[(size_t)3 * B]);
      trunk.I = S3{records[(size_t)4 * B], records[(size_t)5 * B], records[(size_t)6 * B],
                   records[(size_t)7 * B], records[(size_t)8 * B], records[(size_t)9 * B]};
    } else {
      trunk.m = M.mass[0];
      trunk.c = v3(M.com[0][0], M.com[0][1], M.com[0][2]);
      trunk.I = S3{M.inertia[0][0], M.inertia[0][1], M.inertia[0][2], M.inertia[0][3], M.inertia[0][4], M.inertia[0][5]};
    }
  }

  // With the state, in ONE memory round trip (see step_kernel): the DONE word
  // and this step's action (or the fused agent's previous observation).
  float done_word = MODE != MODE_RESET ? SW(UPKIE_S_DONE) : 0.f;
  asm volatile("" : "+v"(done_word));  // pins the load here: the compiler would sink it into the branch that tests it
  float act0 = 0.f, act1 = 0.f;
  float4 prev_obs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == MODE_PENDULUM) {
    if (act) act0 = act[e];
  } else if (fused_agent(MODE)) {
    const float* prev = act ? act : obs;
    prev_obs = reinterpret_cast<const float4*>(prev)[packed ? 2 * (size_t)e : (size_t)e];
  } else if (MODE == MODE_GYROPOD) {
    if (act) {
      const float2 a = reinterpret_cast<const float2*>(act)[e];
      act0 = a.x;
      act1 = a.y;
    }
  } else if (MODE == MODE_BASE_VELOCITY) {
    act0 = bv.mpc_fused ? mpc_velocity[e & 31] : bv.commanded[e];
    act1 = act[2 * (size_t)e + 1];
  }

  // Several env.step() per launch (MODE_PENDULUM_ROLLOUT = the fused agent writing packed records: the
  // agent needs nothing from the host between steps): the state stays in
  // registers from one step to the next, `records` advances by [B][8] per step.
  // What the steps read back from the state words they write is carried in
  // registers too: DONE, the episode / step / elapsed counters.
  constexpr bool ROLLOUT = MODE == MODE_PENDULUM_ROLLOUT;
  int steps_left = ROLLOUT && packed && n_steps > 1 ? n_steps : 1;
  float* records_out = obs;
  float episode_word = ROLLOUT ? SW(UPKIE_S_EPISODE) : 0.f;
  float elapsed_word = ROLLOUT && C.max_episode_steps > 0 ? SW(UPKIE_S_ELAPSED) : 0.f;
  const bool any_noise = C.any_control_noise || C.any_measurement_noise;
  unsigned step_count = any_noise ? (unsigned)SW(UPKIE_S_STEP) : 0u;

  SweepWarmStart sweep_warm_start;  // spans the substeps of ONE env.step() (dynamics.hpp): several steps in a launch = as many launches
  // several steps in a launch: the per-step state words go to memory once, behind the last step, from the registers that
  // mirror them (step_kernel_octet: same memory image as one launch per step, no store address live across the step loop)
  bool reset_seen = false, step_seen = false;
  float tau_stepped[3] = {0.f, 0.f, 0.f};
  // (the prologue's loads land before the step loop: left pending they are waited for inside it, by instructions that
  // from the second step on wait for the record stores of the step before -- step_kernel_octet)
  if (ROLLOUT) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
next_step:
  sweep_warm_start.swept = 0;
  bool do_reset;
  if (MODE == MODE_RESET) {
    do_reset = mask ? mask[e] != 0 : true;
  } else {
    do_reset = (C.autoreset_mode == UPKIE_AUTORESET_NEXT_STEP || C.autoreset_mode == AUTORESET_DONE_PASS) && done_word != 0.f;
    if (C.autoreset_mode == AUTORESET_DONE_PASS) {
      // SAME_STEP autoreset, second launch (see step_kernel): both lanes of a pair agree on DONE
      if (final_obs) {  // every env, see step_kernel
        constexpr int W = ObsWords<MODE>::value;
        if (MODE == MODE_SERVOS) {  // each lane keeps its three servos
#pragma unroll
          for (int i = 0; i < 15; ++i) final_obs[(size_t)30 * e + 15 * leg + i] = obs[(size_t)30 * e + 15 * leg + i];
        } else if (lead) {
          const float* last = obs + (size_t)(packed ? 8 : W) * e;
#pragma unroll
          for (int i = 0; i < W; ++i) final_obs[(size_t)W * e + i] = last[i];
        }
      }
      if (!do_reset) return;
    }
  }
  const float signed_radius = M.left_sign * M.wheel_radius;

  // Gyropod observation from the pair (upkie_gyropod.py:186-214)
  auto observe6 = [&](float yaw_, float yawvel_, float (&o6)[6]) {
    const float qp = xchg(s.q[2]), qdp = xchg(s.qd[2]);
    const float ql = pick(leg, s.q[2], qp), qr = pick(leg, qp, s.q[2]);
    const float qdl = pick(leg, s.qd[2], qdp), qdr = pick(leg, qdp, s.qd[2]);
    float r01 = 2.f * (s.qx * s.qy - s.qz * s.qw), r11 = 1.f - 2.f * (s.qx * s.qx + s.qz * s.qz), r21 = 2.f * (s.qy * s.qz + s.qx * s.qw);
    float x = fminf(fmaxf(2.f * (s.qw * s.qy - s.qz * s.qx), -1.f), 1.f);
    o6[0] = 0.5f * (ql - qr) * signed_radius;
    o6[1] = asinf(x);
    o6[2] = yaw_;
    o6[3] = 0.5f * (qdl - qdr) * signed_radius;
    o6[4] = r01 * s.angvel.x + r11 * s.angvel.y + r21 * s.angvel.z;
    o6[5] = yawvel_;
  };

  if (MODE == MODE_RESET && !do_reset) {
    if (obs) {
      float o6[6];
      observe6(SW(UPKIE_S_YAW), SW(UPKIE_S_YAWVEL), o6);
      if (lead) {
#pragma unroll
        for (int i = 0; i < 6; ++i) obs[(size_t)6 * e + i] = o6[i];
      }
    }
    return;
  }

  // ---- action map (the owned hip, knee and wheel) --------------------------
  Servo cmd[3];
  float a0 = 0.f, a1 = 0.f;
  unsigned episode = 0;
  if (do_reset) {
    episode = ROLLOUT ? (unsigned)episode_word : (unsigned)SW(UPKIE_S_EPISODE);
    Phys full;
    sample_init_state(C, (unsigned)e, episode, full);  // same draws in both lanes
    s.pos = full.pos; s.qw = full.qw; s.qx = full.qx; s.qy = full.qy; s.qz = full.qz;
    s.linvel = full.linvel; s.angvel = full.angvel;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      s.q[k] = pick(leg, full.q[k], full.q[3 + k]);
      s.qd[k] = 0.f;
      cmd[k] = Servo{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
  } else if (MODE == MODE_SERVOS) {
    const float* a = act + (size_t)36 * e + 18 * leg;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float eff = PL.effort[k], vel = PL.velocity[k];
      cmd[k].position = clamp_ref(a[6 * k + 0], PL.lower[k], PL.upper[k]);
      cmd[k].velocity = clamp_ref(a[6 * k + 1], -vel, vel);
      cmd[k].feedforward_torque = clamp_ref(a[6 * k + 2], -eff, eff);
      cmd[k].kp_scale = clamp_ref(a[6 * k + 3], 0.f, C.max_gain_scale);
      cmd[k].kd_scale = clamp_ref(a[6 * k + 4], 0.f, C.max_gain_scale);
      cmd[k].maximum_torque = clamp_ref(a[6 * k + 5], 0.f, eff);
      if (const int replaced = guard_servo_command(cmd[k], eff)) guard_count(C.guard, 0, replaced);  // non-finite guard (step_kernels.hpp)
    }
  } else if (MODE != MODE_RESET) {
    if (fused_agent(MODE)) {
      const float4 o = prev_obs;
      a0 = C.agent_gains[0] * o.x + C.agent_gains[1] * o.y + C.agent_gains[2] * o.z + C.agent_gains[3] * o.w;
      a0 = clamp_ref(a0, -C.agent_clip, C.agent_clip);
    } else {
      a0 = act0;
      a1 = act1;
    }
    {
      const int replaced = guard_velocity_actions(a0, a1, C.max_yaw_velocity);
      if (lead && replaced) guard_count(C.guard, 0, replaced);
    }
    float v = clamp_ref(a0, -C.max_ground_velocity, C.max_ground_velocity);
    float yawd = clamp_ref(a1, -C.max_yaw_velocity, C.max_yaw_velocity);
    float inv_radius = fast_rcp(M.wheel_radius);
    float wheel_velocity = v * inv_radius;
    float left = M.left_sign * wheel_velocity, right = -M.left_sign * wheel_velocity;
    float yaw_to_wheel = M.left_sign * (0.5f * M.wheel_base) * inv_radius;
    left = fmaf(yaw_to_wheel, yawd, left);
    right = fmaf(yaw_to_wheel, yawd, right);
    const float alpha = C.dt / 1.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      legref[k] = legref[k] + alpha * (0.f - legref[k]);
      cmd[k].position = clamp_ref(legref[k], PL.lower[k], PL.upper[k]);
      cmd[k].velocity = 0.f;
      cmd[k].feedforward_torque = 0.f;
      cmd[k].kp_scale = clamp_ref(C.leg_gain_scale, 0.f, C.max_gain_scale);
      cmd[k].kd_scale = cmd[k].kp_scale;
      cmd[k].maximum_torque = PL.effort[k];
    }
    cmd[2].position = NAN;
    cmd[2].velocity = clamp_ref(pick(leg, left, right), -PL.velocity[2], PL.velocity[2]);
    cmd[2].feedforward_torque = 0.f;
    cmd[2].kp_scale = 1.f;
    cmd[2].kd_scale = 1.f;
    cmd[2].maximum_torque = PL.effort[2];
  }

  // ---- substeps ------------------------------------------------------------
  float tau[3] = {0.f, 0.f, 0.f};
  bool contact = false;
  const int nsub = do_reset ? 1 : C.nb_substeps;
  for (int sub = 0; sub < C.nb_substeps; ++sub) {
    if (sub >= nsub) break;
    float zn[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (C.any_control_noise && !do_reset) philox_normal6(C, (unsigned)e, step_count, (unsigned)sub, zn);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      tau[k] = joint_torque(s.q[k], s.qd[k], cmd[k], C.kp, C.kd, PL.friction[k], PL.control_noise[k] * pick(leg, zn[k], zn[3 + k]));
    {
      ConstModelPtr mp = (ConstModelPtr)Mp;
      asm volatile("" : "+s"(mp));
      const ExtForces ext_now{do_reset ? nullptr : ext.force, ext.stride, ext.slots};  // reset steps once without external forces
#if defined(UPKIE_AB_PAIR_SWEEPS_ONLY)  // (A/B build: the two-lane Servos kernels without the active-set solve, as until round 6)
      contact = physics_substep_pair(*mp, Lm, PL, leg, s, tau, C.h, RAND ? &trunk : nullptr, ext_now, &sweep_warm_start);
#else  // UpkieServos envs skid and tumble as a matter of course: their instantiations carry the active-set solve (contact_sweeps_warm)
      contact = physics_substep_pair<MODE == MODE_SERVOS>(*mp, Lm, PL, leg, s, tau, C.h, RAND ? &trunk : nullptr, ext_now, &sweep_warm_start);
#endif
    }
    if (SPINE) {
      // one cycle of the spine's observer pipeline: each lane runs the WheelContact estimator of its
      // own wheel, the env-level quantities are own + partner (identical in both lanes)
#define OM(w) spine_state[(size_t)(w) * B + e]
      const int ow = UPKIE_O_WHEEL + 5 * leg;
      float fv = OM(ow), aa = OM(ow + 1), at = OM(ow + 2), in = OM(ow + 3);
      bool ct = OM(ow + 4) != 0.f;
      if (do_reset) { fv = aa = at = in = 0.f; ct = false; }
      wheel_contact_observe(C.spine, tau[2], s.qd[2], fv, aa, at, in, ct);
      OM(ow) = fv; OM(ow + 1) = aa; OM(ow + 2) = at; OM(ow + 3) = in; OM(ow + 4) = ct ? 1.f : 0.f;
      const float sq = pair_sum(tau[0] * tau[0] + tau[1] * tau[1]);
      const float upper = obs_low_pass(do_reset ? 0.f : OM(UPKIE_O_UPPER_LEG_TORQUE), C.spine.leg_alpha, sqrtf(sq));
      const bool fc = pair_any(ct) || upper > C.spine.upper_leg_torque_threshold;
      float op = do_reset ? 0.f : OM(UPKIE_O_ODOMETRY_POSITION), ov = do_reset ? 0.f : OM(UPKIE_O_ODOMETRY_VELOCITY);
      const float sum = pair_sum(ct ? pick(leg, C.spine.signed_radius[0], C.spine.signed_radius[1]) * s.qd[2] : 0.f);
      const float n = pair_sum(ct ? 1.f : 0.f);
      if (fc) {
        ov = n > 0.f ? sum / n : 0.f;
        op += ov * C.h;
      }
      if (lead) {
        OM(UPKIE_O_UPPER_LEG_TORQUE) = upper;
        OM(UPKIE_O_CONTACT) = fc ? 1.f : 0.f;
        OM(UPKIE_O_ODOMETRY_POSITION) = op;
        OM(UPKIE_O_ODOMETRY_VELOCITY) = ov;
      }
#undef OM
    }
  }

  // ---- non-finite guard: the state behind the substeps (step_kernels.hpp) -------
  bool unsound;
  {
    float mag = fabsf(s.q[0]) + fabsf(s.q[1]) + fabsf(s.q[2]) + fabsf(s.qd[0]) + fabsf(s.qd[1]) + fabsf(s.qd[2]);
    mag = pair_sum(mag);
    mag += fabsf(s.pos.x) + fabsf(s.pos.y) + fabsf(s.pos.z) + fabsf(s.qw) + fabsf(s.qx) + fabsf(s.qy) + fabsf(s.qz);
    mag += fabsf(s.linvel.x) + fabsf(s.linvel.y) + fabsf(s.linvel.z) + fabsf(s.angvel.x) + fabsf(s.angvel.y) + fabsf(s.angvel.z);
    if (YAWING) mag += fabsf(yaw);
    unsound = !(mag < 3.0e38f);
  }
  if (unsound) {  // both lanes of the pair agree (the sum is the same in both)
    s.pos = v3(C.init_pos[0], C.init_pos[1], C.init_pos[2]);
    s.qw = C.init_quat[0]; s.qx = C.init_quat[1]; s.qy = C.init_quat[2]; s.qz = C.init_quat[3];
    s.linvel = v3(C.init_linvel[0], C.init_linvel[1], C.init_linvel[2]);
    s.angvel = v3(C.init_angvel[0], C.init_angvel[1], C.init_angvel[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      s.q[k] = pick(leg, C.init_joint[k], C.init_joint[3 + k]);
      s.qd[k] = 0.f;
      tau[k] = 0.f;
    }
    legref[0] = s.q[0];
    legref[1] = s.q[1];
    yaw = 0.f;
    a1 = 0.f;
    contact = false;
    if (lead) guard_count(C.guard, 1, 1);
  }

  // ---- wrapper post-processing ---------------------------------------------
  bool fallen = false, timeout = false;
  float obs6[6];
  if (ROLLOUT) {
    reset_seen = reset_seen || do_reset;
    if (!do_reset) {
      step_seen = true;
#pragma unroll
      for (int k = 0; k < 3; ++k) tau_stepped[k] = tau[k];
    }
  }
  if (do_reset) {
    legref[0] = s.q[0];
    legref[1] = s.q[1];
    yaw = 0.f;
    yawvel = 0.f;
    if (lead && !ROLLOUT) {
      SW(UPKIE_S_YAW) = 0.f;
      SW(UPKIE_S_YAWVEL) = 0.f;
      SW(UPKIE_S_MPC_V) = 0.f;
      SW(UPKIE_S_SE2_X) = 0.f;
      SW(UPKIE_S_SE2_Y) = 0.f;
      SW(UPKIE_S_EPISODE) = (float)((episode + 1u) & UPKIE_COUNTER_MASK);
      SW(UPKIE_S_DONE) = 0.f;
      SW(UPKIE_S_ELAPSED) = 0.f;
    }
    episode_word = (float)((episode + 1u) & UPKIE_COUNTER_MASK);
    done_word = 0.f;
    elapsed_word = 0.f;
    observe6(yaw, yawvel, obs6);
  } else {
    if (YAWING) {
      yaw = fmaf(a1, C.dt, yaw);
      yawvel = a1;
      if (lead) {
        SW(UPKIE_S_YAW) = yaw;
        SW(UPKIE_S_YAWVEL) = yawvel;
      }
    }
    observe6(yaw, yawvel, obs6);
    fallen = unsound;  // (every env kind: the non-finite guard ends the episode)
    if (MODE != MODE_SERVOS) fallen = fallen || fabsf(obs6[1]) > C.fall_pitch;
    if (fallen && lead && !ROLLOUT) SW(UPKIE_S_DONE) = 1.f;
    if (C.max_episode_steps > 0) {  // time limit, see step_kernel
      const float elapsed = (ROLLOUT ? elapsed_word : SW(UPKIE_S_ELAPSED)) + 1.f;
      timeout = elapsed >= (float)C.max_episode_steps && !fallen;
      elapsed_word = elapsed;
      if (lead && !ROLLOUT) {
        SW(UPKIE_S_ELAPSED) = elapsed;
        if (timeout) SW(UPKIE_S_DONE) = 1.f;
      }
    }
    if (fallen || timeout) done_word = 1.f;
    if (!ROLLOUT) {
#pragma unroll
      for (int k = 0; k < 3; ++k) SWI(UPKIE_S_TORQUE + k, 3 * leg, leg3_off) = tau[k];
    }
    if (any_noise) {
      step_count = (step_count + 1u) & UPKIE_COUNTER_MASK;
      if (lead && !ROLLOUT) SW(UPKIE_S_STEP) = (float)step_count;
    }
  }

  // ---- store (last step of the launch) --------------------------------------
  const bool more_steps = steps_left > 1;
  if (more_steps) {
    // next step of this launch: its agent reads this step's observation
    const float4 o4 = make_float4(obs6[1], obs6[0], obs6[4], obs6[3]);
    if (lead) {
      float4* rec = reinterpret_cast<float4*>(records_out) + 2 * (size_t)e;
      rec[0] = o4;
      rec[1] = make_float4(0.f, fallen ? 1.f : 0.f, timeout ? 1.f : 0.f, 0.f);
    }
    prev_obs = o4;
    records_out += (size_t)8 * B;
    steps_left -= 1;
    goto next_step;
  }
  if (ROLLOUT) {  // the per-step words of the steps of this launch (above)
    if (lead) {
      if (reset_seen) {
        SW(UPKIE_S_YAW) = 0.f;
        SW(UPKIE_S_YAWVEL) = 0.f;
        SW(UPKIE_S_MPC_V) = 0.f;
        SW(UPKIE_S_SE2_X) = 0.f;
        SW(UPKIE_S_SE2_Y) = 0.f;
        SW(UPKIE_S_EPISODE) = episode_word;
      }
      SW(UPKIE_S_DONE) = done_word;
      if (reset_seen || C.max_episode_steps > 0) SW(UPKIE_S_ELAPSED) = elapsed_word;
      if (any_noise && step_seen) SW(UPKIE_S_STEP) = (float)step_count;
    }
    if (step_seen) {
#pragma unroll
      for (int k = 0; k < 3; ++k) SWI(UPKIE_S_TORQUE + k, 3 * leg, leg3_off) = tau_stepped[k];
    }
  }
  if (lead) {
    SW(UPKIE_S_POS) = s.pos.x; SW(UPKIE_S_POS + 1) = s.pos.y; SW(UPKIE_S_POS + 2) = s.pos.z;
    SW(UPKIE_S_QUAT) = s.qw; SW(UPKIE_S_QUAT + 1) = s.qx; SW(UPKIE_S_QUAT + 2) = s.qy; SW(UPKIE_S_QUAT + 3) = s.qz;
    SW(UPKIE_S_LINVEL) = s.linvel.x; SW(UPKIE_S_LINVEL + 1) = s.linvel.y; SW(UPKIE_S_LINVEL + 2) = s.linvel.z;
    SW(UPKIE_S_ANGVEL) = s.angvel.x; SW(UPKIE_S_ANGVEL + 1) = s.angvel.y; SW(UPKIE_S_ANGVEL + 2) = s.angvel.z;
    SW(UPKIE_S_CONTACT) = contact ? 1.f : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    SWI(UPKIE_S_Q + k, 3 * leg, leg3_off) = s.q[k];
    SWI(UPKIE_S_QD + k, 3 * leg, leg3_off) = s.qd[k];
  }
  if (MODE != MODE_SERVOS) {
    SWI(UPKIE_S_LEGREF, 2 * leg, leg2_off) = legref[0];
    SWI(UPKIE_S_LEGREF + 1, 2 * leg, leg2_off) = legref[1];
  }

  if (MODE == MODE_RESET) {
    if (obs && lead) {
#pragma unroll
      for (int i = 0; i < 6; ++i) obs[(size_t)6 * e + i] = obs6[i];
    }
    return;
  }
  if (MODE == MODE_SERVOS) {
    // each lane reports its own three servos (upkie_servos.py:288-306)
    float* o = obs + (size_t)30 * e + 15 * leg;
    float zm[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (C.any_measurement_noise) philox_normal6(C, (unsigned)e, step_count, NOISE_SLOT_MEASUREMENT, zm);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      o[5 * k + 0] = s.q[k];
      o[5 * k + 1] = s.qd[k];
      o[5 * k + 2] = (do_reset ? SWI(UPKIE_S_TORQUE + k, 3 * leg, leg3_off) : tau[k]) + PL.measurement_noise[k] * pick(leg, zm[k], zm[3 + k]);
      o[5 * k + 3] = 42.0f;
      o[5 * k + 4] = 18.0f;
    }
  }
  if (!lead) return;
  if (MODE == MODE_PENDULUM || fused_agent(MODE)) {
    const float4 o4 = make_float4(obs6[1], obs6[0], obs6[4], obs6[3]);
    if (packed) {
      float4* rec = reinterpret_cast<float4*>(records_out) + 2 * (size_t)e;
      rec[0] = o4;
      if (C.autoreset_mode != AUTORESET_DONE_PASS) rec[1] = make_float4(0.f, fallen ? 1.f : 0.f, timeout ? 1.f : 0.f, 0.f);
      return;
    }
    reinterpret_cast<float4*>(obs)[e] = o4;
  } else if (MODE == MODE_BASE_VELOCITY) {
    float x = 0.f, y = 0.f;
    if (!do_reset) {
      float lin = act[2 * (size_t)e];
      if (!is_finite(lin)) lin = 0.f;  // (non-finite guard)
      float sy, cy;
      sincosf(yaw, &sy, &cy);
      x = fmaf(lin * cy, C.dt, SW(UPKIE_S_SE2_X));
      y = fmaf(lin * sy, C.dt, SW(UPKIE_S_SE2_Y));
      SW(UPKIE_S_SE2_X) = x;
      SW(UPKIE_S_SE2_Y) = y;
    }
    obs[(size_t)3 * e] = x;
    obs[(size_t)3 * e + 1] = y;
    obs[(size_t)3 * e + 2] = yaw;
    reinterpret_cast<float4*>(bv.x0)[e] = make_float4(obs6[0], obs6[1], obs6[3], obs6[4]);
    bv.contact[e] = contact ? 1 : 0;
  } else if (MODE == MODE_GYROPOD) {
    float2* o2 = reinterpret_cast<float2*>(obs) + (size_t)3 * e;
    o2[0] = make_float2(obs6[0], obs6[1]);
    o2[1] = make_float2(obs6[2], obs6[3]);
    o2[2] = make_float2(obs6[4], obs6[5]);
  }
  if (C.autoreset_mode == AUTORESET_DONE_PASS) return;  // reward and flags are those of the terminal step
  reward[e] = 0.f;
  terminated[e] = fallen ? 1 : 0;
  truncated[e] = timeout ? 1 : 0;
#undef SW
#undef SWI
}
