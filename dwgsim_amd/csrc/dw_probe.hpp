// dw_probe.hpp -- the measurement hooks of the kernels, PRODUCT FORM: every one of them is nothing.
// The kernels name the places where an analysis build may look in (phase clocks) or switch a part off (to weigh it); what that means is defined
// by the header of this name that comes first on the include path.  The product build finds this file: the hooks are constant-false predicates
// and empty statements, so the compiled code holds no trace of them.  The analysis twin is tools/probe/dw_probe.hpp (tools/knockout_build.sh,
// tools/phase_profile.sh put -Itools/probe in front); it is never part of a library that ships or that the tests load.
#pragma once

namespace dw { namespace probe {
// "is part `bit` of the kernel switched off in this build?" -- never, in the product
constexpr bool off(int) { return false; }
// values an analysis build keeps alive after it removed their consumer
template <class... T> __device__ __forceinline__ void keep(const T &...) {}
} }
// phase clocks of k_simulate (shader-clock ticks per phase, added to the batch's counters by the analysis twin)
#define DW_PROBE_INIT() do { } while (0)
#define DW_PROBE_MARK(args, k) do { } while (0)
#define DW_PROBE_MARKF(args, k) do { } while (0)      // (the finer marks of the -DDW_PHASE_FINE analysis build)
