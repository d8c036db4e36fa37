// In-library kernel timing with HIP events on the launch stream (bench.py's live roofline numbers).
#pragma once
#include <hip/hip_runtime.h>

#define DCN_PROF_TAGS 56
// tags: 0-2 igemm NT tiles 128x128 / 128x64 / 256x32, 3-4 igemm NN tiles 128x128 / 128x64,
//       5 wgrad / TN GEMM (all tiles), 6-7 igemm 64x128 NT / NN tiles, 8 l2norm+score fwd, 9 l2norm+score bwd, 10 scale_act,
//       11 bn backward apply, 12 exp+sums (co-attention), 13/14 latency-bound small GEMMs (< 1024 rows: LSTM steps),
//       15 igemm 128x128 NT with the 32-float K-step, 16 igemm split-bf16 128x128 NT, 17 wgrad / TN split-bf16 128x128, 18 igemm split-bf16 256x64 NT, 19 igemm bf16-operand tiles, 20 wgrad / TN bf16-operand tiles, 21 igemm split-bf16 128x128 NN,
//       22 BatchNorm backward reduce (channel_partials), 23 igemm fp8-operand tiles, 24-27 f16-split igemm 128x128 / wgrad / 256x64 / NN,
//       30 reduce_slabs (split-K slabs of the weight gradient), 31 dA (co-attention backward, elementwise over E),
//       32 wgrad3 (3x3 stride-1 weight gradient by filter rows, f16 split),
//       28 / 29 conv3 (3x3 stride-1 strip kernel, f16 split): 256x128 / 128x128 tile, 33 conv3 with bf16 operands,
//       34 stem (direct 3x3 conv of the 4-channel image, HBM-priced), 35 conv1 (1x1 layers, both tiles by LDS-DMA, f16 split)
int prof_begin(int tag, double work, hipStream_t s, double bytes = 0.0);   // returns a record id or -1 (disabled / full); bytes: algorithmic HBM bytes of a FLOP-priced launch
void prof_end(int id, hipStream_t s);
