// loop_step_kernel.h -- loop.h's step as a kernel of its own: behind an in-library RCCL all-reduce, for the estimators
// whose reduction does not take the step itself, and to re-open a stepping loop.  Included by mi_icp.hip only.
#pragma once
#include "loop.h"

namespace mi {

static __global__ __launch_bounds__(kStepThreads) void loop_step_kernel(DevLoop* st_g, double* sys_in, int resume, MailArgs mail) {
    __shared__ DevLoop st_s;
    loop_step_block(st_g, sys_in, resume, st_s, StepPre{false, 0u, 0.0, 0u}, mail);
}

}  // namespace mi
