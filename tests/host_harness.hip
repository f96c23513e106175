// Test-only: runs the product's device functions on the HOST so that a single
// physics substep can be compared with the oracle without a GPU.
#include "../upkie_amd/csrc/upkie_hip.hip"

extern "C" int harness_substep(const UpkieModel* model, float* st, const float* tau, float h, const float* records,
                               const float* ext_forces, const UpkieExternalForces* ext_slots) {
  DevModel M;
  std::string why;
  if (!convert_model(model, &M, &why)) return -1;
  Phys s;
  s.pos = v3(st[UPKIE_S_POS], st[UPKIE_S_POS + 1], st[UPKIE_S_POS + 2]);
  s.qw = st[UPKIE_S_QUAT]; s.qx = st[UPKIE_S_QUAT + 1]; s.qy = st[UPKIE_S_QUAT + 2]; s.qz = st[UPKIE_S_QUAT + 3];
  s.linvel = v3(st[UPKIE_S_LINVEL], st[UPKIE_S_LINVEL + 1], st[UPKIE_S_LINVEL + 2]);
  s.angvel = v3(st[UPKIE_S_ANGVEL], st[UPKIE_S_ANGVEL + 1], st[UPKIE_S_ANGVEL + 2]);
  for (int j = 0; j < 6; ++j) { s.q[j] = st[UPKIE_S_Q + j]; s.qd[j] = st[UPKIE_S_QD + j]; }
  float t[6];
  for (int j = 0; j < 6; ++j) t[j] = tau[j];
  ExtSlots x{};
  if (ext_forces && ext_slots) {
    x.count = ext_slots->count;
    for (int i = 0; i < x.count; ++i) {
      x.body[i] = ext_slots->body[i];
      x.local[i] = ext_slots->local[i];
      for (int k = 0; k < 3; ++k) x.point[i][k] = (float)ext_slots->point[i][k];
    }
  }
  const ExtForces ext{ext_forces && ext_slots ? ext_forces : nullptr, 1, &x};  // [count][3], one env
  DevLimits Lm;
  model_limits(M, &Lm);
  BodyInertials bi;
  if (records) load_body_inertials(records, 1, bi);  // [70], one env
  bool c = physics_substep(M, Lm, s, t, h, records ? &bi : nullptr, ext);
  st[UPKIE_S_POS] = s.pos.x; st[UPKIE_S_POS + 1] = s.pos.y; st[UPKIE_S_POS + 2] = s.pos.z;
  st[UPKIE_S_QUAT] = s.qw; st[UPKIE_S_QUAT + 1] = s.qx; st[UPKIE_S_QUAT + 2] = s.qy; st[UPKIE_S_QUAT + 3] = s.qz;
  st[UPKIE_S_LINVEL] = s.linvel.x; st[UPKIE_S_LINVEL + 1] = s.linvel.y; st[UPKIE_S_LINVEL + 2] = s.linvel.z;
  st[UPKIE_S_ANGVEL] = s.angvel.x; st[UPKIE_S_ANGVEL + 1] = s.angvel.y; st[UPKIE_S_ANGVEL + 2] = s.angvel.z;
  for (int j = 0; j < 6; ++j) { st[UPKIE_S_Q + j] = s.q[j]; st[UPKIE_S_QD + j] = s.qd[j]; }
  return c ? 1 : 0;
}

// fuse_links() on the host: link factors [UPKIE_MAX_LINKS] -> records [70]
extern "C" int harness_fuse_links(const UpkieModel* model, const float* factors, float* records) {
  DevLinks L;
  std::string why;
  if (!convert_links(model, &L, &why)) return -1;
  float f[UPKIE_MAX_LINKS];
  for (int l = 0; l < UPKIE_MAX_LINKS; ++l) f[l] = factors[l];
  fuse_links(L, f, records, 1);
  return L.count;
}
