// Device kernels of the gfx950 SpGEMM engine (row-wise Gustavson, C = A * B, fp64).
//
// Execution model used throughout: 64-lane wavefronts; a "group" is G consecutive lanes
// (G in {8,16,32,64}) inside one wavefront.  LDS operations of one wavefront execute in issue
// order, so lanes of a wavefront hand data to each other through LDS with a compiler-only
// fence (lds_fence) and no s_barrier.
//
// Ordering rule that makes the result bit-identical to the reference's CPU loop
// (src/mults/drop_parallel.cpp:46-57, accumulator include/matrix_impl.hpp:81-95): every output
// row is owned by ONE group; the group walks the A row's entries sequentially in storage order
// (the k order of the CPU loop) and, for one k, its lanes take different entries of B row k,
// which have distinct columns.  Every (i,j) therefore receives its additions in the same order
// as on the CPU, products and sums are rounded separately (no fused multiply-add), and no
// two lanes of one instruction ever add to the same accumulator.
#pragma once

// The kernels, by theme:
#include "kernels_common.h"
#include "kernels_prep.h"
#include "kernels_walk.h"
#include "kernels_symbolic.h"
#include "kernels_numeric.h"
#include "kernels_big.h"
#include "kernels_short.h"
#include "kernels_util.h"
