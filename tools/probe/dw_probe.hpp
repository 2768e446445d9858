// tools/probe/dw_probe.hpp -- ANALYSIS ONLY: the twin of dwgsim_amd/csrc/dw_probe.hpp with the hooks switched on.
//   -DDW_KNOCK=<bits>   parts of k_simulate / the flow model switched off, to weigh them (output is garbage by construction, only the kernel
//                       time means something): 1 no text assembly at all, 2 text assembled but not stored, 4 no error tests, 8 no base
//                       extraction, 16 no header, 64 producers kept alive but no assembly, 256 no first draws of the flow model, 1024 no
//                       pass 2 of the flow model, 4096 first draws made but none scores; further bits: tools/knockout_build.sh
//   -DDW_PHASE_TIMING   per wave, shader-clock ticks spent in each phase of k_simulate are added to counters[8 + phase]
// Found first by the builds of tools/knockout_build.sh and tools/phase_profile.sh (-Itools/probe); never by the product build or the tests.
#pragma once
#ifndef DW_KNOCK
#define DW_KNOCK 0
#endif

namespace dw { namespace probe {
constexpr bool off(int bit) { return (DW_KNOCK & bit) != 0; }
__device__ __forceinline__ void keep() {}
template <class T, class... R> __device__ __forceinline__ void keep(const T &v, const R &... r) { asm volatile("" :: "v"(v)); keep(r...); }
} }
#ifdef DW_PHASE_TIMING
#define DW_PROBE_INIT() uint64_t ph_t = __builtin_amdgcn_s_memtime()
#define DW_PROBE_MARK_(args, k) do { const uint64_t ph_n = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&(args).counters[8 + (k)], (unsigned long long)(ph_n - ph_t)); ph_t = ph_n; } while (0)
#ifdef DW_PHASE_FINE
// the fine split of what happens in front of the text: the coarse marks 3 .. 6 (name lengths + look-backs 2-3, header, sequence, quality) are ONE phase (4), mark 2 becomes 3 = the
// barrier behind the error phase (the wait for wave 0's look-back of the random-read index), and the fine marks take 5 = the last attempt's extraction + leaving the attempt loop,
// 6 = the block scan of the random reads (a barrier: the block's slowest wave), 2 = the error phase's own work; 1 is then what wave 0 spends in look-back 1 (nothing in the others)
#define DW_PROBE_MARK(args, k) DW_PROBE_MARK_(args, ((k) == 2 ? 3 : (k) >= 3 && (k) <= 6 ? 4 : (k)))
#define DW_PROBE_MARKF(args, k) DW_PROBE_MARK_(args, k)
#else
#define DW_PROBE_MARK(args, k) DW_PROBE_MARK_(args, k)
#define DW_PROBE_MARKF(args, k) do { } while (0)
#endif
#else
#define DW_PROBE_INIT() do { } while (0)
#define DW_PROBE_MARK(args, k) do { } while (0)
#define DW_PROBE_MARKF(args, k) do { } while (0)
#endif
