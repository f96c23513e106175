// step_instances.hip -- definitions of one group of step-kernel instantiations (step_instances.hpp):
//   hipcc -c -DUPKIE_INSTANCE_GROUP=<0..15> step_instances.hip
// or, for a look at ONE eight-lane kernel's ISA in seconds (tools/isa_probe.sh):
//   hipcc -S --cuda-device-only -DUPKIE_PROBE_OCTET_MODE=<Mode> [-DUPKIE_PROBE_RAND=true] [-DUPKIE_PROBE_DEFAULT_SCALARS=true]
//         [-DUPKIE_PROBE_IN_PLACE=true] step_instances.hip
#define UPKIE_STEP_INSTANCES_ONLY 1
#if defined(UPKIE_PROBE_OCTET_MODE)
#include "step_kernels.hpp"
#if !defined(UPKIE_PROBE_RAND)
#define UPKIE_PROBE_RAND false
#endif
#if !defined(UPKIE_PROBE_DEFAULT_SCALARS)
#define UPKIE_PROBE_DEFAULT_SCALARS false
#endif
#if !defined(UPKIE_PROBE_IN_PLACE)
#define UPKIE_PROBE_IN_PLACE false
#endif
#if !defined(UPKIE_PROBE_BULLET_LIKE)
#define UPKIE_PROBE_BULLET_LIKE false
#endif
template __global__ void upkie::step_kernel_octet<UPKIE_PROBE_OCTET_MODE, UPKIE_PROBE_RAND, UPKIE_PROBE_DEFAULT_SCALARS, UPKIE_PROBE_IN_PLACE, UPKIE_PROBE_BULLET_LIKE>(
    const upkie::DevModel*, const upkie::DevParams*, int, int, float*, const float*, float*, float*, uint8_t*, uint8_t*, const uint8_t*,
    const float*, const float*, int, upkie::BaseVelocityPtrs, float*, int, unsigned*, upkie::ServoPolicyArg<UPKIE_PROBE_OCTET_MODE>, float*);
#else
#if !defined(UPKIE_INSTANCE_GROUP)
#error "compile with -DUPKIE_INSTANCE_GROUP=<0..15> (upkie_amd/lib.py builds every group)"
#endif
#define UPKIE_INSTANCE_KW
#include "step_instances.hpp"
#endif
