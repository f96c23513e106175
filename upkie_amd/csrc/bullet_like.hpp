// bullet_like.hpp -- the Bullet-like contact specification on the device, selectable per handle next to the product's
// default one (upkie_sim_set_contact_manifold, include/upkie_hip.h; `contact_model="bullet_like"` of the vector envs).
//
// What pybullet.stepSimulation() (call site pybullet_backend.py:306; reset: :228) is published to do between the free
// acceleration and the position update, as SURVEY.md Appendix B.1 / B.2 summarise Bullet 3.25's btPersistentManifold
// and btMultiBodyConstraintSolver [third party, absent from /root/reference: restated, unverified]:
//   - a PERSISTENT manifold of up to four points per tire: every step the cached points are refreshed (dropped beyond
//     the contact breaking threshold, along the normal or in the plane), and the deepest point of the tire replaces
//     the cached point nearest to it in the wheel's frame within the threshold, or is added (a full manifold gives up
//     its shallowest point);
//   - one normal and two friction rows per cached point; friction directions along / across the sliding velocity of
//     the point (btPlaneSpace1 of the normal when it does not slide); zero friction CFM; the normal rows carry the
//     URDF contact stiffness / damping as CFM / ERP and may only close a gap within the step;
//   - sequential impulses, a FIXED number of sweeps (numSolverIterations = 50, DevModel::pgs_iterations): joint-limit
//     rows, then the normal rows, then the two friction rows of each point together, projected onto the cone
//     |f| <= mu f_n; normal impulses warm-started with 0.85 x the impulse applied in the previous step.
// The same algorithm as the test-side fp64 checker's `bullet_like_contacts` (world frame, dense Delassus matrix),
// stated here the way Bullet itself iterates: on VELOCITIES (each row keeps M^-1 J' and adds its
// impulse change to the running velocity change), in the base frame, on top of the factored system the default
// specification builds (System: 6 x 6 base block + the two 3 x 3 leg blocks). In exact arithmetic the two produce the
// same impulses after every sweep.
//
// One env per lane: this file serves the one-lane step kernels -- every case, every entry point, any batch size
// (launch_step sends a handle with a contact manifold here beyond 16384 envs and for Servos steps). Up to 16384 envs the
// envs whose legs the servos hold run the eight-lane variant of the same algorithm (octet.hpp, oct_bullet_like_solve).
// Rows of the general path live in private memory (up to 28 rows x 29 words, indexed dynamically: scratch); the manifold
// (64 words per env) is loaded from / stored to the caller's buffer [BL_MANIFOLD_WORDS][B] once per env.step().
#pragma once
#include "dynamics.hpp"

namespace upkie {

enum {
  BL_POINTS = 4,
  BL_POINT_WORDS = 8,  // point in the wheel frame (3), on the plane in world coordinates (3), applied normal impulse, live
  BL_MANIFOLD_WORDS = 2 * BL_POINTS * BL_POINT_WORDS,  // == UPKIE_CONTACT_MANIFOLD_WORDS
  BL_ROWS = 2 * BL_POINTS * 3 + UPKIE_NJ  // every cached point's three rows + a limit row for EVERY joint (a model may bound its wheel joints too)
};

#if !defined(__HIP_DEVICE_COMPILE__)
// Host tests only (tests/test_bullet_like_on_host.py): the eight-lane solve (octet.hpp) records here the system it
// sweeps -- W [36], right-hand sides [6], warm start [6], tires on the floor [2] -- and, per sweep, the largest change of
// an impulse. That is the measurement behind "the FIXED sweeps stay fixed" (profiles/r05_bullet_like_sweeps.txt): leaving
// the loop at the exact fixed point, at a short limit cycle or at fp32 resolution does not shorten a wavefront's loop.
struct BulletLikeProbe {
  float system[50];
  float change[64];  // [it]: max |lam after sweep it - before|
  float lam[64][6];  // impulses after sweep it
  int sweeps;
};
inline BulletLikeProbe* g_bullet_like_probe = nullptr;
#endif

// ---- the sweeps of the six-row case, shared by the one-lane and the eight-lane kernels (round 5) --------------------
// Six rows: [normal, friction 1, friction 2] of the left tire's point, then of the right tire's. W: their Delassus
// matrix (symmetric; only entries between live tires are read as couplings), rhs, lam: the warm start in / the impulses
// out (0 on a tire that is off), cfm_n: constraint force mixing added to the two normal rows' diagonal, mu: friction
// coefficient. A FIXED number of Gauss-Seidel sweeps in the published order -- left normal, right normal (which sees
// the left one's new impulse), the left point's friction pair projected onto its cone, the right point's --, each
// update x_r = lam_r + (rhs_r - sum_c W_rc lam_c - cfm lam_r) / (W_rr + cfm), restated with every row divided by its
// diagonal ONCE: x_r = b_r - sum_{c != r} a_rc lam_c (five multiply-adds a row instead of six plus two instructions;
// a tire that is off has zero rows: its impulses stay 0 without a select per update). The two rows a stage updates
// from the same impulses -- a point's friction pair; the two normals up to the one term that couples them -- are
// accumulated as PACKED fp32 pairs (v_pk_fma_f32: the broadcast of an impulse to both halves, and the swap inside the
// pair's own 2 x 2 block, are the instruction's op_sel modifiers, no move). The projection onto the cone is the scale
// min(mu lam_n / |x|, 1) (one v_rsq, no compare / select; |x| = 0 gives inf or NaN, which min() turns into 1).
// About 36 vector instructions a sweep (the row-by-row form both kernels carried until round 4: ~85), the same update rule
// and row order: the fp64 checker's `bullet_like_contacts` remains its twin.
typedef float BlPair __attribute__((ext_vector_type(2)));
UPKIE_HD BlPair bl_fma(BlPair a, BlPair b, BlPair c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_elementwise_fma(a, b, c);
#else
  return BlPair{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)};
#endif
}
UPKIE_HD BlPair bl_lo(BlPair v) { return __builtin_shufflevector(v, v, 0, 0); }
UPKIE_HD BlPair bl_hi(BlPair v) { return __builtin_shufflevector(v, v, 1, 1); }
UPKIE_HD BlPair bl_swap(BlPair v) { return __builtin_shufflevector(v, v, 1, 0); }

UPKIE_HD void bullet_like_sweeps6(const float (&W)[6][6], const float (&rhs)[6], float (&lam)[6], const bool (&on)[2], float cfm_n, float mu,
                                  int iterations) {
  // rows divided by their diagonal, negated (x = b + sum na lam); a tire that is off: zero rows
  float ninv[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) ninv[r] = on[r / 3] ? -fast_rcp(W[r][r] + ((r % 3) == 0 ? cfm_n : 0.f)) : 0.f;
  auto na = [&](int r, int c) { return W[r][c] * ninv[r]; };
  auto b = [&](int r) { return -(rhs[r] * ninv[r]); };
  const BlPair N0 = {na(0, 1), na(3, 1)}, N1 = {na(0, 2), na(3, 2)}, N2 = {na(0, 4), na(3, 4)}, N3 = {na(0, 5), na(3, 5)}, Nb = {b(0), b(3)};
  const float n03 = na(0, 3), n30 = na(3, 0);
  const BlPair L0 = {na(1, 0), na(2, 0)}, L1 = {na(1, 3), na(2, 3)}, L2 = {na(1, 4), na(2, 4)}, L3 = {na(1, 5), na(2, 5)}, L4 = {na(1, 2), na(2, 1)},
               Lb = {b(1), b(2)};
  const BlPair R0 = {na(4, 0), na(5, 0)}, R1 = {na(4, 3), na(5, 3)}, R2 = {na(4, 1), na(5, 1)}, R3 = {na(4, 2), na(5, 2)}, R4 = {na(4, 5), na(5, 4)},
               Rb = {b(4), b(5)};
  BlPair Pn = {lam[0], lam[3]}, PL = {lam[1], lam[2]}, PR = {lam[4], lam[5]};
  // The order of the instructions in the loop is set BY HAND (BL_KEEP_ORDER: a scheduling barrier between statements
  // that are one instruction each; hazard wait states stay the compiler's business). On gfx950 a packed multiply-add that
  // reads the result of the instruction just in front of it costs a wait state -- an issue slot of a lone wavefront, like
  // any instruction: 8.5 instead of 4.9 cycles a link, profiles/r05_pk_rate.txt -- and the scheduler, left alone, lines
  // each accumulation chain up back to back (8 wait states in a 48-slot loop). So every stage keeps two chains in
  // flight, and the loop is ROTATED: the terms that do not wait for this sweep's normal impulses (the friction pairs'
  // own 2 x 2 blocks and their couplings to each other) are accumulated while the normal rows finish, and the part of
  // the NEXT sweep's normal rows that reads the left point's friction impulses as soon as those are final.
#if defined(__HIP_DEVICE_COMPILE__)
#define BL_KEEP_ORDER __builtin_amdgcn_sched_barrier(0)
#else
#define BL_KEEP_ORDER (void)0
#endif
  BlPair p = bl_fma(N1, bl_hi(PL), bl_fma(N0, bl_lo(PL), Nb));
  auto sweep = [&]() {
    // normal rows: the right point's friction impulses; beside them the friction pairs' terms that know no normal impulse
    BlPair q = N2 * bl_lo(PR);              BL_KEEP_ORDER;
    BlPair x = bl_fma(L2, bl_lo(PR), Lb);   BL_KEEP_ORDER;
    q = bl_fma(N3, bl_hi(PR), q);           BL_KEEP_ORDER;
    x = bl_fma(L3, bl_hi(PR), x);           BL_KEEP_ORDER;
    p = p + q;                              BL_KEEP_ORDER;
    x = bl_fma(L4, bl_swap(PL), x);         BL_KEEP_ORDER;
    BlPair y = bl_fma(R4, bl_swap(PR), Rb); BL_KEEP_ORDER;
    // ... and the term that chains the two normal rows: left, then right with the left one's new impulse
    const float l0 = fmaxf(fmaf(n03, Pn.y, p.x), 0.f);
    const float l3 = fmaxf(fmaf(n30, l0, p.y), 0.f);
    Pn = BlPair{l0, l3};                    BL_KEEP_ORDER;
    // the friction pairs: their coupling to the new normal impulses
    x = bl_fma(L0, bl_lo(Pn), x);           BL_KEEP_ORDER;
    y = bl_fma(R0, bl_lo(Pn), y);           BL_KEEP_ORDER;
    x = bl_fma(L1, bl_hi(Pn), x);           BL_KEEP_ORDER;
    y = bl_fma(R1, bl_hi(Pn), y);           BL_KEEP_ORDER;
    const float lim_left = mu * Pn.x;       BL_KEEP_ORDER;
    {  // the left point's pair projected onto the cone |f| <= mu f_n
      const float rs = fast_rsqrt(fmaf(x.x, x.x, x.y * x.y));  BL_KEEP_ORDER;
      const float lim_right = mu * Pn.y;    BL_KEEP_ORDER;  // (an independent instruction behind the v_rsq: its result is not read by the next one)
      const float scale = fminf(lim_left * rs, 1.f);
      PL = x * BlPair{scale, scale};        BL_KEEP_ORDER;
      // the right point's pair sees the left one's new impulses -- and so do the next sweep's normal rows
      p = bl_fma(N0, bl_lo(PL), Nb);        BL_KEEP_ORDER;
      y = bl_fma(R2, bl_lo(PL), y);         BL_KEEP_ORDER;
      p = bl_fma(N1, bl_hi(PL), p);         BL_KEEP_ORDER;
      y = bl_fma(R3, bl_hi(PL), y);         BL_KEEP_ORDER;
      const float scale_right = fminf(lim_right * fast_rsqrt(fmaf(y.x, y.x, y.y * y.y)), 1.f);
      PR = y * BlPair{scale_right, scale_right};  BL_KEEP_ORDER;
    }
  };
  // (five sweeps per trip -- the published 50 are ten trips --: the impulses ping-pong between register pairs instead of
  // being copied back, and the loop's own instructions -- counter, compare, branch, the copies a loop-carried pair costs
  // -- are paid once per five sweeps)
  int it = 0;
  for (; it + 4 < iterations; it += 5) {
    sweep();
    sweep();
    sweep();
    sweep();
    sweep();
  }
  for (; it < iterations; ++it) sweep();
#undef BL_KEEP_ORDER
  lam[0] = Pn.x; lam[3] = Pn.y;
  lam[1] = PL.x; lam[2] = PL.y;
  lam[4] = PR.x; lam[5] = PR.y;
}

struct BlRow {
  float Jb[6];   // base part of the row (base frame: linear 0-2, angular 3-5)
  float Jl[3];   // joint part: the three joints of leg `leg` (the other leg's entries are zero)
  float Mb[6], Ml[3], Mr[3];  // M^-1 J'
  float rhs, cfm, inv_diag, lam;
  int leg, kind, normal_row, slot;  // kind: 0 normal, 1 friction, 2 joint limit; slot: manifold word of the applied impulse or -1
};

// Contacts and joint limits of one substep under the Bullet-like specification.
//   S: the factored system of this substep; bf: its base frame; tb / tl / tr: on entry the generalised impulse
//   h (applied - bias) (base, left leg, right leg), on return the velocity change of the substep, contacts included;
//   mf: the env's contact manifold (updated). Returns the floor-contact flag (a cached point exists).
template <class ModelT>
UPKIE_HD bool bullet_like_contacts(const ModelT& M, const DevLimits& Lm, const System& S, const BaseFrame& bf, const Phys& s, float h,
                                   float (&tb)[6], float (&tl)[3], float (&tr)[3], float (&mf)[BL_MANIFOLD_WORDS], ContactReport* report) {
  // (report: PyBulletBackend.get_contact_points for this substep -- per tire whether a point is cached, its (first) point
  // in base coordinates and the force, in base coordinates, its points' impulses sum to)
  if (report) {
    report->active[0] = report->active[1] = false;
    report->force[0] = report->force[1] = v3(0.f, 0.f, 0.f);
  }
  // free velocity: nu + M^-1 h (applied - bias)
  system_solve<true, true>(S, tb, tl, tr);
  const V3 vF = bf.vB + v3(tb[0], tb[1], tb[2]), wF = bf.wB + v3(tb[3], tb[4], tb[5]);
  float qdF[UPKIE_NJ];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    qdF[j] = s.qd[j] + tl[j];
    qdF[3 + j] = s.qd[3 + j] + tr[j];
  }
  const V3 nB = bf.nB;
  const float breaking = M.contact_breaking_threshold, ih = fast_rcp(h);
  const float denom = h * M.contact_stiffness + M.contact_damping;
  const float erp = denom > 0.f ? h * M.contact_stiffness * fast_rcp(denom) : 0.2f;
  const float cfm_n = denom > 0.f ? fast_rcp(denom * h) : 0.f;

  BlRow rows[BL_ROWS];
  int nrows = 0;
  bool any_contact = false;
  auto finish_row = [&](BlRow& R) {  // M^-1 J' and the diagonal of the Delassus matrix
#pragma unroll
    for (int c = 0; c < 6; ++c) R.Mb[c] = R.Jb[c];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      R.Ml[j] = R.leg == 0 ? R.Jl[j] : 0.f;
      R.Mr[j] = R.leg == 0 ? 0.f : R.Jl[j];
    }
    system_solve<true, true>(S, R.Mb, R.Ml, R.Mr);
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 6; ++c) d = fmaf(R.Jb[c], R.Mb[c], d);
#pragma unroll
    for (int j = 0; j < 3; ++j) d = fmaf(R.Jl[j], R.leg == 0 ? R.Ml[j] : R.Mr[j], d);
    R.inv_diag = 1.f / (d + R.cfm);
  };

  // ---- joint limits (btMultiBodyJointLimitConstraint, ERP 0.2): first in every sweep
  if (Lm.enforce) {
    for (int j = 0; j < UPKIE_NJ; ++j) {
      float bias;  // (the gap-aware row of joint_limit_row, dynamics.hpp: the same rule under both contact models)
      const float sign = joint_limit_row(Lm.bounded[j] != 0, s.q[j], Lm.lower[j], Lm.upper[j], joint_limit_reach(s.qd[j], M.max_joint_velocity, h), ih, bias);
      if (sign == 0.f) continue;
      BlRow& R = rows[nrows];
#pragma unroll
      for (int c = 0; c < 6; ++c) R.Jb[c] = 0.f;
      R.leg = j / 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) R.Jl[k] = (j % 3) == k ? sign : 0.f;
      R.kind = 2; R.normal_row = nrows; R.cfm = 0.f; R.lam = 0.f; R.slot = -1;
      R.rhs = -sign * qdF[j] + bias;
      finish_row(R);
      ++nrows;
    }
  }

  // ---- the tires: orientation of each wheel body in the base frame (one rotation about y by the leg's summed joint
  // angles), then the manifold bookkeeping
  float wsn[2], wcs[2];
#pragma unroll
  for (int wheel = 0; wheel < 2; ++wheel) {
    const Leg& G = S.leg[wheel];
    joint_sincos(G.sgn[0] * s.q[3 * wheel] + G.sgn[1] * s.q[3 * wheel + 1] + G.sgn[2] * s.q[3 * wheel + 2], &wsn[wheel], &wcs[wheel]);
  }
  auto world_xy = [&](V3 A, float& x, float& y) {
    x = s.pos.x + bf.r00 * A.x + bf.r01 * A.y + bf.r02 * A.z;
    y = s.pos.y + bf.r10 * A.x + bf.r11 * A.y + bf.r12 * A.z;
  };
  // the three rows (normal, two friction directions) of one cached point `pt` of `wheel`
  auto point_rows = [&](int wheel, const float* pt, int applied_slot, BlRow& Rn, BlRow& R1, BlRow& R2, int first_row) {
    const Leg& G = S.leg[wheel];
    const V3 A = G.o[2] + rot_y(wcs[wheel], wsn[wheel], v3(pt[0], pt[1], pt[2]));
    const float dist = s.pos.z + dot(nB, A);
    // velocity of the point at the free velocity
    V3 v = vF + cross(wF, A);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const V3 rr = A - G.o[j];
      const float sq = G.sgn[j] * (wheel == 0 ? qdF[j] : qdF[3 + j]);
      v = v + sq * v3(rr.z, 0.f, -rr.x);  // (sgn y) x rr
    }
    const float vn = dot(v, nB);
    const V3 vt = v - vn * nB;
    const float lat2 = dot(vt, vt);
    V3 t1, t2;
    if (lat2 > 1.1920929e-07f) {  // SIMD_EPSILON: friction along the sliding direction
      t1 = (1.f / sqrtf(lat2)) * vt;
      t2 = cross(t1, nB);
    } else {  // btPlaneSpace1(n) for n = world z: (0, -1, 0) and (1, 0, 0), in base coordinates
      t1 = v3(-bf.r10, -bf.r11, -bf.r12);
      t2 = v3(bf.r00, bf.r01, bf.r02);
    }
    auto one = [&](BlRow& R, V3 d, int k) {
      const V3 Axd = cross(A, d);
      R.Jb[0] = d.x; R.Jb[1] = d.y; R.Jb[2] = d.z; R.Jb[3] = Axd.x; R.Jb[4] = Axd.y; R.Jb[5] = Axd.z;
      R.leg = wheel;
      float rel = d.x * vF.x + d.y * vF.y + d.z * vF.z + Axd.x * wF.x + Axd.y * wF.y + Axd.z * wF.z;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const V3 rr = A - G.o[j];
        R.Jl[j] = G.sgn[j] * (rr.z * d.x - rr.x * d.z);
        rel = fmaf(R.Jl[j], wheel == 0 ? qdF[j] : qdF[3 + j], rel);
      }
      if (k == 0) {
        R.kind = 0; R.normal_row = first_row; R.cfm = cfm_n;
        R.rhs = dist <= 0.f ? -rel + erp * (-dist) * ih : -rel - dist * ih;
        R.lam = 0.85f * pt[6];  // m_warmstartingFactor
        R.slot = applied_slot;
      } else {
        R.kind = 1; R.normal_row = first_row; R.cfm = 0.f;
        R.rhs = -rel;
        R.lam = 0.f;
        R.slot = -1;
      }
      finish_row(R);
    };
    one(Rn, nB, 0);
    one(R1, t1, 1);
    one(R2, t2, 2);
  };
  int live_points[2] = {0, 0}, live_slot[2] = {0, 0};
  for (int wheel = 0; wheel < 2; ++wheel) {
    const Leg& G = S.leg[wheel];
    float* pts = mf + wheel * BL_POINTS * BL_POINT_WORDS;
    const V3 ow = G.o[2];
    const float sn = wsn[wheel], cs = wcs[wheel];
    auto in_base = [&](const float* pt) { return ow + rot_y(cs, sn, v3(pt[0], pt[1], pt[2])); };
    // btPersistentManifold::refreshContactPoints
    for (int p = 0; p < BL_POINTS; ++p) {
      float* pt = pts + p * BL_POINT_WORDS;
      if (pt[7] == 0.f) continue;
      const V3 A = in_base(pt);
      const float dist = s.pos.z + dot(nB, A);
      float x, y;
      world_xy(A, x, y);
      const float dx = x - pt[3], dy = y - pt[4];
      if (dist > breaking || dx * dx + dy * dy > breaking * breaking) pt[7] = 0.f;
    }
    // the deepest point of the tire circle (btConvexPlaneCollisionAlgorithm's support point)
    {
      const float un = fast_sqrt(nB.x * nB.x + nB.z * nB.z);
      if (un >= 1e-6f) {
        const float iun = fast_rcp(un);
        const V3 center = ow + v3(M.wheel_center[wheel][0], M.wheel_center[wheel][1], M.wheel_center[wheel][2]);
        const V3 P = center + M.wheel_radius * v3(-nB.x * iun, 0.f, -nB.z * iun);
        const float Pz = s.pos.z + dot(nB, P);
        if (Pz <= breaking) {
          const V3 local = rot_y(cs, -sn, P - ow);
          // btPersistentManifold::getCacheEntry: the nearest cached point within the threshold is replaced
          int slot = -1;
          float nearest = breaking * breaking;
          for (int p = 0; p < BL_POINTS; ++p) {
            const float* pt = pts + p * BL_POINT_WORDS;
            if (pt[7] == 0.f) continue;
            const float d0 = pt[0] - local.x, d1 = pt[1] - local.y, d2 = pt[2] - local.z;
            const float dd = d0 * d0 + d1 * d1 + d2 * d2;
            if (dd < nearest) { nearest = dd; slot = p; }
          }
          float keep = 0.f;
          if (slot >= 0) {
            keep = pts[slot * BL_POINT_WORDS + 6];  // replaceContactPoint keeps the applied impulse
          } else {
            for (int p = 0; p < BL_POINTS && slot < 0; ++p)
              if (pts[p * BL_POINT_WORDS + 7] == 0.f) slot = p;
            if (slot < 0) {  // full: the shallowest cached point makes room
              float worst = -3.0e38f;
              for (int p = 0; p < BL_POINTS; ++p) {
                const float z = s.pos.z + dot(nB, in_base(pts + p * BL_POINT_WORDS));
                if (z > worst) { worst = z; slot = p; }
              }
            }
          }
          float* pt = pts + slot * BL_POINT_WORDS;
          pt[0] = local.x; pt[1] = local.y; pt[2] = local.z;
          world_xy(P, pt[3], pt[4]);
          pt[5] = 0.f;
          pt[6] = keep;
          pt[7] = 1.f;
        }
      }
    }
    for (int p = 0; p < BL_POINTS; ++p)
      if (pts[p * BL_POINT_WORDS + 7] != 0.f) {
        live_points[wheel] += 1;
        live_slot[wheel] = p;
        any_contact = true;
      }
  }
  // The common case -- at most ONE cached point per tire (a wheel that rolls: its deepest point replaces the cached
  // one) and no joint at its stop -- has a fixed layout of six rows: rows, Delassus matrix and impulses stay in
  // registers and the sweeps are the dense Gauss-Seidel of the published algorithm, row for row. Anything else (several
  // points on a tire, limit rows) goes through the general row list below, in private memory.
  const bool fast = nrows == 0 && live_points[0] <= 1 && live_points[1] <= 1;
  if (fast) {
    if (!any_contact) return false;  // (tb, tl, tr hold the free velocity change)
    BlRow F[6];
    bool on[2];
#pragma unroll
    for (int wheel = 0; wheel < 2; ++wheel) {
      on[wheel] = live_points[wheel] == 1;
      const int word = (wheel * BL_POINTS + live_slot[wheel]) * BL_POINT_WORDS;
      float pt[BL_POINT_WORDS];
#pragma unroll
      for (int i = 0; i < BL_POINT_WORDS; ++i) pt[i] = on[wheel] ? mf[word + i] : 0.f;
      point_rows(wheel, pt, word + 6, F[3 * wheel], F[3 * wheel + 1], F[3 * wheel + 2], 3 * wheel);
    }
    float W[6][6], lam[6], rhs[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        float w = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) w = fmaf(F[a].Jb[c], F[b].Mb[c], w);
#pragma unroll
        for (int j = 0; j < 3; ++j) w = fmaf(F[a].Jl[j], a < 3 ? F[b].Ml[j] : F[b].Mr[j], w);
        W[a][b] = on[a / 3] && on[b / 3] ? w : 0.f;
      }
      lam[a] = on[a / 3] ? F[a].lam : 0.f;
      rhs[a] = F[a].rhs;
    }
    bullet_like_sweeps6(W, rhs, lam, on, cfm_n, M.friction_mu, M.pgs_iterations);
#pragma unroll
    for (int w = 0; w < 2; ++w)
      if (on[w]) mf[(w * BL_POINTS + live_slot[w]) * BL_POINT_WORDS + 6] = lam[3 * w];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = 0; c < 6; ++c) tb[c] = fmaf(F[a].Mb[c], lam[a], tb[c]);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        tl[j] = fmaf(F[a].Ml[j], lam[a], tl[j]);
        tr[j] = fmaf(F[a].Mr[j], lam[a], tr[j]);
      }
    }
    if (report) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        if (!on[w]) continue;
        const float* pt = mf + (w * BL_POINTS + live_slot[w]) * BL_POINT_WORDS;
        report->active[w] = true;
        report->point[w] = S.leg[w].o[2] + rot_y(wcs[w], wsn[w], v3(pt[0], pt[1], pt[2]));
#pragma unroll
        for (int k = 0; k < 3; ++k)  // (the linear part of a contact row is its direction)
          report->force[w] = report->force[w] + (ih * lam[3 * w + k]) * v3(F[3 * w + k].Jb[0], F[3 * w + k].Jb[1], F[3 * w + k].Jb[2]);
      }
    }
    return true;
  }
  for (int wheel = 0; wheel < 2; ++wheel) {
    float* pts = mf + wheel * BL_POINTS * BL_POINT_WORDS;
    // rows of every cached point
    {
      for (int p = 0; p < BL_POINTS; ++p) {
        float* pt = pts + p * BL_POINT_WORDS;
        if (pt[7] == 0.f) continue;
        point_rows(wheel, pt, (wheel * BL_POINTS + p) * BL_POINT_WORDS + 6, rows[nrows], rows[nrows + 1], rows[nrows + 2], nrows);
        nrows += 3;
      }
    }
  }
  if (nrows == 0) return false;  // (tb, tl, tr hold the free velocity change)

  // ---- sequential impulses on the velocity change dv = M^-1 J' lam
  float dvb[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dvl[3] = {0.f, 0.f, 0.f}, dvr[3] = {0.f, 0.f, 0.f};
  auto apply = [&](const BlRow& R, float delta) {
#pragma unroll
    for (int c = 0; c < 6; ++c) dvb[c] = fmaf(R.Mb[c], delta, dvb[c]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dvl[j] = fmaf(R.Ml[j], delta, dvl[j]);
      dvr[j] = fmaf(R.Mr[j], delta, dvr[j]);
    }
  };
  auto along = [&](const BlRow& R) {  // J dv
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < 6; ++c) a = fmaf(R.Jb[c], dvb[c], a);
#pragma unroll
    for (int j = 0; j < 3; ++j) a = fmaf(R.Jl[j], R.leg == 0 ? dvl[j] : dvr[j], a);
    return a;
  };
  for (int r = 0; r < nrows; ++r)
    if (rows[r].lam != 0.f) apply(rows[r], rows[r].lam);  // the warm start
  const float mu = M.friction_mu;
  for (int it = 0; it < M.pgs_iterations; ++it) {
    for (int pass = 0; pass < 2; ++pass) {  // joint limits, then normals
      for (int r = 0; r < nrows; ++r) {
        BlRow& R = rows[r];
        if (R.kind != (pass == 0 ? 2 : 0)) continue;
        float x = R.lam + (R.rhs - along(R) - R.cfm * R.lam) * R.inv_diag;
        x = x < 0.f ? 0.f : x;
        const float delta = x - R.lam;
        R.lam = x;
        apply(R, delta);
      }
    }
    for (int r = 0; r + 1 < nrows; ++r) {  // the two friction rows of a point together, projected onto the cone
      BlRow& R1 = rows[r];
      if (R1.kind != 1 || r != R1.normal_row + 1) continue;
      BlRow& R2 = rows[r + 1];
      const float w1 = along(R1), w2 = along(R2);
      float x1 = R1.lam + (R1.rhs - w1) * R1.inv_diag, x2 = R2.lam + (R2.rhs - w2) * R2.inv_diag;
      const float lim = mu * rows[R1.normal_row].lam, n2 = x1 * x1 + x2 * x2;
      if (n2 > lim * lim) {
        const float sc = lim * fast_rsqrt(n2);
        x1 *= sc;
        x2 *= sc;
      }
      const float d1 = x1 - R1.lam, d2 = x2 - R2.lam;
      R1.lam = x1;
      R2.lam = x2;
      apply(R1, d1);
      apply(R2, d2);
    }
  }
  for (int r = 0; r < nrows; ++r)
    if (rows[r].slot >= 0) mf[rows[r].slot] = rows[r].lam;
  if (report) {
    for (int w = 0; w < 2; ++w)
      for (int p = BL_POINTS - 1; p >= 0; --p) {  // (the first live point of the tire ends up reported)
        const float* pt = mf + (w * BL_POINTS + p) * BL_POINT_WORDS;
        if (pt[7] == 0.f) continue;
        report->active[w] = true;
        report->point[w] = S.leg[w].o[2] + rot_y(wcs[w], wsn[w], v3(pt[0], pt[1], pt[2]));
      }
    for (int r = 0; r < nrows; ++r) {
      const BlRow& R = rows[r];
      if (R.kind == 2) continue;
      const V3 f = (ih * R.lam) * v3(R.Jb[0], R.Jb[1], R.Jb[2]);
      if (R.leg == 0) report->force[0] = report->force[0] + f; else report->force[1] = report->force[1] + f;
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) tb[c] += dvb[c];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    tl[j] += dvl[j];
    tr[j] += dvr[j];
  }
  return any_contact;
}

}  // namespace upkie
