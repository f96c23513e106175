// step_instances.hpp -- every instantiation of the three step kernels that launch_step (upkie_hip.hip) can launch, as
// one list read twice: upkie_hip.hip includes it with UPKIE_INSTANCE_KW = `extern` (declarations: the C-ABI's
// translation unit compiles no step kernel), step_instances.hip with UPKIE_INSTANCE_KW empty and UPKIE_INSTANCE_GROUP = g
// (definitions of group g). ~140 kernels of 5-20 k instructions each: in one translation unit the library took two
// minutes to build; by sixteen groups, on eight cores, about thirty seconds (upkie_amd/lib.py).
//
// The kernels of different groups share no device symbol (every device function is inlined), so no relocatable device
// code is needed: each object carries its own code object, the host-side launch stubs are ordinary weak symbols.
#pragma once
#include "step_kernels.hpp"

#if !defined(UPKIE_INSTANCE_KW)
#define UPKIE_INSTANCE_KW extern
#define UPKIE_INSTANCE_GROUP (-1) /* declarations: every group */
#endif
#define UPKIE_INSTANCE_GROUPS 16

namespace upkie {

#define UPKIE_ONE_LANE_ARGS                                                                                                             \
  const DevModel*, DevLimits, DevConfig, float*, const float*, float*, float*, uint8_t*, uint8_t*, const uint8_t*, const float*, \
      const float*, int, BaseVelocityPtrs, float*, float*
#define UPKIE_PAIR_ARGS UPKIE_ONE_LANE_ARGS, int
#define UPKIE_ONE_LANE_KERNEL_ARGS UPKIE_ONE_LANE_ARGS, float*
#define UPKIE_OCTET_ARGS(MODE)                                                                                                          \
  const DevModel*, const DevParams*, int, int, float*, const float*, float*, float*, uint8_t*, uint8_t*, const uint8_t*, const float*, \
      const float*, int, BaseVelocityPtrs, float*, int, unsigned*, ServoPolicyArg<MODE>, float*

// one env per lane: <MODE, RAND, waves per SIMD, in-step spine observers>
#define UPKIE_ONE_LANE(MODE)                                                                                      \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, false, 1, false, false>(UPKIE_ONE_LANE_KERNEL_ARGS); \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, false, 1, true, false>(UPKIE_ONE_LANE_KERNEL_ARGS);  \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, false, 2, false, false>(UPKIE_ONE_LANE_KERNEL_ARGS); \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, false, 2, true, false>(UPKIE_ONE_LANE_KERNEL_ARGS);  \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, true, 1, false, false>(UPKIE_ONE_LANE_KERNEL_ARGS);  \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, true, 1, true, false>(UPKIE_ONE_LANE_KERNEL_ARGS);   \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, true, 2, false, false>(UPKIE_ONE_LANE_KERNEL_ARGS);  \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, true, 2, true, false>(UPKIE_ONE_LANE_KERNEL_ARGS);
// one env per lane under the Bullet-like contact model (upkie_sim_set_contact_manifold): <MODE, RAND, 1, false, true>
#define UPKIE_ONE_LANE_BULLET(MODE)                                                                                \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, false, 1, false, true>(UPKIE_ONE_LANE_KERNEL_ARGS); \
  UPKIE_INSTANCE_KW template __global__ void step_kernel<MODE, true, 1, false, true>(UPKIE_ONE_LANE_KERNEL_ARGS);
// two lanes per env: <MODE, RAND, in-step spine observers>
#define UPKIE_PAIR(MODE)                                                                          \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_pair<MODE, false, false>(UPKIE_PAIR_ARGS); \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_pair<MODE, false, true>(UPKIE_PAIR_ARGS);  \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_pair<MODE, true, false>(UPKIE_PAIR_ARGS);  \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_pair<MODE, true, true>(UPKIE_PAIR_ARGS);
// eight lanes per env: <MODE, RAND, default model's scalars as constants, SAME_STEP autoreset inside the launch>
#define UPKIE_OCTET(MODE, D, IP)                                                                                       \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_octet<MODE, false, D, IP, false>(UPKIE_OCTET_ARGS(MODE)); \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_octet<MODE, true, D, IP, false>(UPKIE_OCTET_ARGS(MODE));
// eight lanes per env under the Bullet-like contact model: <MODE, RAND, false, false, true>
#define UPKIE_OCTET_BULLET(MODE)                                                                                         \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_octet<MODE, false, false, false, true>(UPKIE_OCTET_ARGS(MODE)); \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_octet<MODE, true, false, false, true>(UPKIE_OCTET_ARGS(MODE));


// ... with the SAME_STEP autoreset inside the launch (the modes that reset in place): <MODE, RAND, false, true, true>
#define UPKIE_OCTET_BULLET_IN_PLACE(MODE)                                                                               \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_octet<MODE, false, false, true, true>(UPKIE_OCTET_ARGS(MODE)); \
  UPKIE_INSTANCE_KW template __global__ void step_kernel_octet<MODE, true, false, true, true>(UPKIE_OCTET_ARGS(MODE));

#define UPKIE_IN_GROUP(g) (UPKIE_INSTANCE_GROUP < 0 || UPKIE_INSTANCE_GROUP == (g))

// (sixteen groups, the one-lane kernels -- the largest and slowest to compile -- one mode per group: on eight cores the
// build is as long as its total work allows, not as long as its largest group; lib.build starts the heavy groups first)
#if UPKIE_IN_GROUP(0)
UPKIE_ONE_LANE(MODE_PENDULUM)
#endif
#if UPKIE_IN_GROUP(1)
UPKIE_ONE_LANE(MODE_PENDULUM_AGENT)
#endif
#if UPKIE_IN_GROUP(2)
UPKIE_ONE_LANE(MODE_GYROPOD)
#endif
#if UPKIE_IN_GROUP(3)
UPKIE_ONE_LANE(MODE_BASE_VELOCITY)
#endif
#if UPKIE_IN_GROUP(4)
UPKIE_ONE_LANE(MODE_SERVOS)
#endif
#if UPKIE_IN_GROUP(5)
UPKIE_ONE_LANE(MODE_RESET)
UPKIE_PAIR(MODE_RESET)
UPKIE_OCTET(MODE_RESET, false, false)
UPKIE_OCTET_BULLET_IN_PLACE(MODE_SERVOS)
#endif
#if UPKIE_IN_GROUP(6)
UPKIE_ONE_LANE_BULLET(MODE_RESET)
UPKIE_ONE_LANE_BULLET(MODE_PENDULUM)
UPKIE_ONE_LANE_BULLET(MODE_PENDULUM_AGENT)
#endif
#if UPKIE_IN_GROUP(7)
UPKIE_ONE_LANE_BULLET(MODE_GYROPOD)
UPKIE_ONE_LANE_BULLET(MODE_SERVOS)
UPKIE_ONE_LANE_BULLET(MODE_BASE_VELOCITY)
#endif
#if UPKIE_IN_GROUP(8)
UPKIE_PAIR(MODE_PENDULUM)
UPKIE_PAIR(MODE_PENDULUM_AGENT)
UPKIE_PAIR(MODE_PENDULUM_ROLLOUT)
#endif
#if UPKIE_IN_GROUP(9)
UPKIE_PAIR(MODE_GYROPOD)
UPKIE_PAIR(MODE_SERVOS)
UPKIE_PAIR(MODE_BASE_VELOCITY)
#endif
#if UPKIE_IN_GROUP(10)
UPKIE_OCTET(MODE_PENDULUM, false, false)
UPKIE_OCTET(MODE_PENDULUM, true, false)
UPKIE_OCTET(MODE_PENDULUM, false, true)
UPKIE_OCTET(MODE_PENDULUM, true, true)
#endif
#if UPKIE_IN_GROUP(11)
UPKIE_OCTET(MODE_PENDULUM_AGENT, false, false)
UPKIE_OCTET(MODE_PENDULUM_AGENT, true, false)
UPKIE_OCTET(MODE_PENDULUM_ROLLOUT, false, false)
UPKIE_OCTET(MODE_PENDULUM_ROLLOUT, true, false)
#endif
#if UPKIE_IN_GROUP(12)
UPKIE_OCTET(MODE_GYROPOD, false, false)
UPKIE_OCTET(MODE_GYROPOD, true, false)
UPKIE_OCTET(MODE_GYROPOD, false, true)
UPKIE_OCTET(MODE_GYROPOD, true, true)
#endif
#if UPKIE_IN_GROUP(13)
UPKIE_OCTET(MODE_SERVOS, false, false)
UPKIE_OCTET(MODE_SERVOS, false, true)
UPKIE_OCTET(MODE_BASE_VELOCITY, false, false)
UPKIE_OCTET_BULLET_IN_PLACE(MODE_GYROPOD)
#endif
#if UPKIE_IN_GROUP(14)
UPKIE_OCTET_BULLET(MODE_RESET)
UPKIE_OCTET_BULLET(MODE_PENDULUM)
UPKIE_OCTET_BULLET(MODE_PENDULUM_AGENT)
UPKIE_OCTET_BULLET(MODE_PENDULUM_ROLLOUT)
UPKIE_OCTET_BULLET_IN_PLACE(MODE_PENDULUM)
#endif
#if UPKIE_IN_GROUP(15)
UPKIE_OCTET_BULLET(MODE_GYROPOD)
UPKIE_OCTET_BULLET(MODE_BASE_VELOCITY)
UPKIE_OCTET_BULLET(MODE_SERVOS)
#endif

}  // namespace upkie
