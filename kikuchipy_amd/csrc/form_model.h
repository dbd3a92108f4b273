// form_model.h - constants of the cost model that picks the f32 match kernel of a sweep (api.hip: decide_form,
// choose_nsplit).  Units: the time of one 128-pattern dictionary tile of match.hip against one 256-pattern row block.
// Fitted on profiles/r03_form_choice.json and r03_form_choice_run3.json (tools/form_probe.py: both kernels forced over N = 6 250 .. 300 000, M = 512 ..
// 40 000, K = 2819 / 3600 / 14 400 on one MI355X); tests/test_gpu_engine.py re-measures a sub-grid and fails when the
// automatic choice is more than 3 % behind the better kernel.  Round 4 (profiles/r04_form_choice.json, 80 shapes, after the
// wide kernel's first tile became 0.04 ms cheaper and launches of 29 row blocks kept their XCD grid): worst point 2.9 %
// (10 000 x 50 000 x 120^2 -> wide), mean 0.08 %; FORM_WIDE_LAUNCH = 1.0 / 1.05 re-measured on the whole grid: worst 3.1 /
// 3.0 %, mean 0.16 % - the constants stay.
#pragma once
namespace kpdi {
// PLANNING (choose_nsplit): cost factor of a plan whose dictionary splits are not a multiple of 8 - such a launch has no XCD
// grid (plan_xcd_grid) and its workgroups of one row block are spread over all XCDs.  Round 2 priced that at 1.02; forcing
// multiples of 8 measured 3 - 7 % FASTER steps wherever a plan changed (M = 10 000 / 40 000: 40 / 157 row blocks, e.g.
// 10 000 x 37 500 x 60^2 on match.hip 21.4 -> 20.5 ms with 8 splits x 32 row blocks x 2 launches instead of 6 x 40 x 1;
// 40 000 x 12 500 on the wide kernel 30.6 -> 28.4 ms).
constexpr double FORM_ODD_SPLIT_CLASSIC = 1.25;
constexpr double FORM_ODD_SPLIT_WIDE = 1.2;
// CHOICE (decide_form): a wide plan that still ends on such a split count is this much slower than its tile count says
constexpr double FORM_WIDE_ODD = 1.12;
// match.hip: fixed cost of a launch, and of its quarter-tile tail launch
constexpr double FORM_CLASSIC_LAUNCH = 0.25;
constexpr double FORM_CLASSIC_TAIL = 0.25;
// match16.hip's f32 form: a 256-pattern tile costs 2 / FORM_WIDE_GAIN units; fixed cost of a launch (the first tile's
// candidates go through the buffers, the lists are built at the end)
// FORM_WIDE_GAIN(K) = FORM_WIDE_GAIN + FORM_WIDE_GAIN_K (1 - 3600 / K): the per-tile work outside the MFMA loop (epilogue,
// list handling) is amortised over more steps at large K - measured wide / classic 0.99 at K = 3600, 0.97 at K = 14 400
constexpr double FORM_WIDE_GAIN = 1.035;
constexpr double FORM_WIDE_GAIN_K = 0.02;
// (KPDI_FORM_WIDE_LAUNCH overrides it for fitting runs of tools/form_probe.py; after the first tile's candidates left the
// buffers - launch 0.20 -> 0.166 ms - 1.0 and 0.9 leave the grid's worst point and mean regret where they are, 0.8 adds a 7 % miss)
constexpr double FORM_WIDE_LAUNCH = 1.1;
// Round 6 (profiles/r06_form_choice_launch0.6.json / _launch0.45.json, 80 shapes, with the last partial round on tailgemm.hip): for sweeps of ONE launch
// (M <= 4096 here) the wide kernel now wins from 12 500 dictionary patterns up - one rank's share of configs[1] at N = 8
// 2.92 vs 3.01 ms, at N = 4 5.57 vs 5.67 - which the constant above (fitted when a launch cost 0.2 ms and the tail a quarter
// round) hid; 0.6 gets every single-launch shape of the grid right (worst 0.6 %).  Sweeps of several launches (M = 10 000:
// 32 + 8 row blocks) keep 1.1: with 0.6 the grid's worst point was 14 % (10 000 x 37 500, wide chosen, its second launch
// a quarter full).
constexpr double FORM_WIDE_LAUNCH_SINGLE = 0.6;
// ... its partial units (halves / quarters of a tile) cost this much more per row than whole tiles
constexpr double FORM_WIDE_HALF = 1.1;
constexpr double FORM_WIDE_QUARTER = 1.25;
// tailgemm.hip in place of the partial units: one round of its 32 x 128 workgroups over the chip, and the fixed cost of its
// two launches (the GEMM and the select pass), in tile-times of the wide kernel (first guess; fitted below)
constexpr double FORM_TAIL_GEMM_UNIT = 0.08;
constexpr double FORM_TAIL_GEMM_LAUNCH = 0.02;
}  // namespace kpdi
