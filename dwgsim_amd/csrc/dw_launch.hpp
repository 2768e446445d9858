// dw_launch.hpp -- host-callable launchers of the kernels in dw_walk.hip and dw_simulate.hip
#pragma once
#include <hip/hip_runtime.h>
#include "dw_kernels.hpp"

namespace dw {
void launch_pack(hipStream_t st, const uint8_t *ascii, uint8_t *ref, uint8_t *h0, uint8_t *h1, int64_t l);
void launch_site_scan_list(hipStream_t st, const uint8_t *refview, int64_t l, SegTab seg, WalkParams wp, uint64_t *status, uint64_t *ticket, int32_t *out, uint32_t cap, uint64_t *n_out);
uint32_t site_scan_blocks(int64_t l);            // blocks of the site scan over l positions, and the positions a block takes
uint32_t site_scan_block_positions();
void launch_site_scan_slots(hipStream_t st, const uint8_t *refview, int64_t l, SegTab seg, WalkParams wp, int32_t *slots, uint32_t slot_cap, uint32_t *aux, int32_t *out, uint32_t cap, uint64_t *n_out, uint32_t *over);
void launch_mark_dirty(hipStream_t st, const Event *ev, Count n, const int32_t *lo, uint32_t *dirty);
void launch_dirty_chunks(hipStream_t st, bool restore, const uint32_t *dirty, uint32_t n_words, int64_t l_live, const uint8_t *ref, const uint8_t *refview, const uint16_t *refsumm, const uint16_t *refsumm2,
                         uint8_t *cells0, uint8_t *cells1, uint8_t *view0, uint8_t *view1, uint16_t *summ0, uint16_t *summ1, uint16_t *summ2_0, uint16_t *summ2_1);
void launch_scan_excl(hipStream_t st, uint32_t *data, uint32_t n, uint64_t *total_out);
void launch_compact(hipStream_t st, const uint16_t *mask, const uint32_t *block_base, int32_t *out, int64_t l, uint32_t cap);
void launch_events(hipStream_t st, const int32_t *cand, Count n, const uint8_t *ref, SegTab seg, WalkParams wp, Event *ev, uint32_t *max_del);
void walk_debug_seg_min(uint32_t rows);      // test hook: candidate capacity from which k_scan4 / k_sufmin run segmented (0 = default)
void launch_resolve(hipStream_t st, Event *ev, Count n, const uint32_t *max_del, uint4 *flags, uint32_t *tot4);
void launch_apply(hipStream_t st, Event *ev, Count n, const uint4 *flags, ContigDev c, WalkParams wp);
void launch_justify_seq(hipStream_t st, const Event *ev, Count n, ContigDev c);
void launch_justify(hipStream_t st, const Event *ev, Count n, ContigDev c, int32_t *lo, int32_t *sufmin, uint8_t *bound);
void launch_mut_debug(hipStream_t st, const uint8_t *ref, const uint8_t *h0, const uint8_t *h1, int64_t l, uint64_t *verdict);
void launch_make_view(hipStream_t st, const uint8_t *cells0, const uint8_t *cells1, int64_t n_cells, int64_t l_live, uint8_t *view0, uint8_t *view1, uint16_t *summ0, uint16_t *summ1, uint16_t *summ2_0, uint16_t *summ2_1);
void launch_apply_patches(hipStream_t st, const int32_t *pos, const uint16_t *cells, uint32_t n, uint8_t *h0, uint8_t *h1);
void launch_collect_mask(hipStream_t st, const uint8_t *h0, const uint8_t *h1, int64_t l, uint16_t *mask, uint32_t *block_count);
void launch_gather(hipStream_t st, const int32_t *pos, uint32_t n, const uint8_t *ref, const uint8_t *h0, const uint8_t *h1, uint32_t *cells);
void launch_place(hipStream_t st, const SimArgs &a);
void launch_simulate(hipStream_t st, const SimArgs &a);
void launch_calibrate(hipStream_t st, const CalibArgs &a);
void launch_failrule(hipStream_t st, const uint32_t *meta, uint64_t n_pairs, uint32_t opens_contig, uint64_t *summ, uint64_t *counters, uint64_t *chain);
void launch_chain_set(hipStream_t st, uint64_t *chain, uint64_t rand_base, int set_rand, uint64_t carry, int set_carry);
void launch_init(hipStream_t st, uint64_t *counters, uint32_t n_counters, uint64_t *z0, uint64_t n0, uint64_t *z1, uint64_t n1, uint64_t *chain, uint64_t rand_base, int set_rand, uint64_t carry, int set_carry);
void launch_selftest_lazy(hipStream_t st, int mode, uint32_t first, uint64_t n, double sigma, float qk, float qeps, float qlmin, int qnear1, uint64_t *out);
uint64_t gz_chunks(uint64_t n);
uint64_t gz_capacity(uint64_t n);
void gz_host_tables(uint32_t *crc_table, uint32_t *crc_shift);
void launch_gzip(hipStream_t st, const uint8_t *text, const uint64_t *n_dev, uint64_t text_cap, uint8_t *out, uint64_t out_cap, uint64_t *status, uint64_t *ticket, uint64_t *total, uint64_t *flags,
                 const uint32_t *crc_table, const uint32_t *crc_shift);
void launch_count_byte(hipStream_t st, const uint8_t *text, uint64_t n, uint32_t byte, uint64_t *out);
void launch_selftest_fp64(hipStream_t st, uint32_t seed, uint64_t n, uint64_t *mism);
void launch_selftest_text(hipStream_t st, uint64_t first, uint64_t n, uint64_t stride, uint64_t *out);
}
