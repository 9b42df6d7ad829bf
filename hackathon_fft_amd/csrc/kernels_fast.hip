// kernels_fast.hip -- configuration table + launchers of the register-butterfly Stockham tile
// kernels (template: tile_kernel.h) for MI355X (gfx950).
#include <cstdlib>
#include <cstring>

#include "fast_table.h"
#include "mifft_config.h"

namespace mifft {

static const FastEntry kFastTable[] = {
    // Variants chosen with tools/tune/tune_tile.hip on MI355X (min-of-5 interleaved rounds; numbers in
    // DESIGN.md).  Forcing more waves/SIMD than the butterflies' live registers allow spills and
    // loses 2-3x, so MINW is only raised where the kernel fits.
    // ---- contiguous dimension, fp32 ----
    // 1024: LDS twiddles at 4 waves/SIMD.  (Register twiddles + prefetch measured the same 0.299 ms but sit on
    // the 256-VGPR edge: a refactor that added 20 B of scratch doubled their time.)
    // streaming twin: four small-radix passes, 512 threads x 8 elements (low VGPR count, 32 waves/CU): 0.274 ms
    MIFFT_CFG_STREAM("rows1024_4x4x8x8", float, MIFFT_F32, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM_R("rows1024_16x8x8", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
    MIFFT_CFG_MID_ST("rows1024_16x8x8", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
    MIFFT_CFG_MID_ST_R("rows1024_16x8x8", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
    MIFFT_CFG_CR("rows1024_16x8x8", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
    // power-of-two rows (tools/tune GROUP 9-12, 819-MB tensors): plain 0.304-0.316 ms, streaming twins 0.277-0.295 ms
    MIFFT_CFG_STREAM("rows512_4x4x4x8", float, MIFFT_F32, 512, 4, 4, 4, 4, 8, 8, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM_R("rows512_8x8x8", float, MIFFT_F32, 512, 3, 8, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_MID_ST("rows512_8x8x8", float, MIFFT_F32, 512, 3, 8, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_MID_ST_R("rows512_8x8x8", float, MIFFT_F32, 512, 3, 8, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_CR("rows512_8x8x8", float, MIFFT_F32, 512, 3, 8, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM("rows256_8x8x4", float, MIFFT_F32, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM_R("rows256_8x8x4", float, MIFFT_F32, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_MID_ST("rows256_8x8x4", float, MIFFT_F32, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_MID_ST_R("rows256_8x8x4", float, MIFFT_F32, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_CR("rows256_8x8x4", float, MIFFT_F32, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM("rows128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
    MIFFT_CFG_STREAM_R("rows128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
    MIFFT_CFG_MID_ST("rows128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
    MIFFT_CFG_MID_ST_R("rows128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
    MIFFT_CFG_CR("rows128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
    MIFFT_CFG_MID_ST("rows64_4x4x4", float, MIFFT_F32, 64, 3, 4, 4, 4, 1, 32, 256, false, true, true, TW_REG, 1, false),
    MIFFT_CFG_MID_ST_R("rows64_4x4x4", float, MIFFT_F32, 64, 3, 4, 4, 4, 1, 32, 256, false, true, true, TW_REG, 1, false),
    MIFFT_CFG_CR("rows64_4x4x4", float, MIFFT_F32, 64, 3, 4, 4, 4, 1, 32, 256, false, true, true, TW_REG, 1, false),
    MIFFT_CFG_STREAM("rows2048_4x8x8x8", float, MIFFT_F32, 2048, 4, 4, 8, 8, 8, 2, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM_R("rows2048_16x16x8", float, MIFFT_F32, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_MID_ST("rows2048_16x16x8", float, MIFFT_F32, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_MID_ST_R("rows2048_16x16x8", float, MIFFT_F32, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_CR("rows2048_16x16x8", float, MIFFT_F32, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM("rows4096_8x8x8x8", float, MIFFT_F32, 4096, 4, 8, 8, 8, 8, 1, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM_R("rows4096_16x16x16", float, MIFFT_F32, 4096, 3, 16, 16, 16, 1, 1, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_MID_ST("rows4096_16x16x16", float, MIFFT_F32, 4096, 3, 16, 16, 16, 1, 1, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_MID_ST_R("rows4096_16x16x16", float, MIFFT_F32, 4096, 3, 16, 16, 16, 1, 1, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_CR("rows4096_16x16x16", float, MIFFT_F32, 4096, 3, 16, 16, 16, 1, 1, 256, false, true, true, TW_LDS, 2, false),
    // one 128-KiB row per workgroup; twiddles from the global table (the compact LDS table would need 131 KB more)
    MIFFT_CFG("rows16384_16x16x8x8", float, MIFFT_F32, 16384, 4, 16, 16, 8, 8, 1, 1024, false, true, true, TW_GLOBAL, 4, false),
    MIFFT_CFG_STREAM_ST("rows93_31x3", float, MIFFT_F32, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
    MIFFT_CFG_STREAM_ST_R("rows93_31x3", float, MIFFT_F32, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
    MIFFT_CFG_SMALL_ST("rows93_31x3", float, MIFFT_F32, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
    MIFFT_CFG_SMALL_ST_R("rows93_31x3", float, MIFFT_F32, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
    MIFFT_CFG_CR("rows93_31x3", float, MIFFT_F32, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
#if defined(MIFFT_ALT_ROWS480) && MIFFT_ALT_ROWS480 == 1   // A/B builds only (tools/ab_lib.sh)
    MIFFT_CFG_NTL("rows480_6x8x10_t8x512", float, MIFFT_F32, 480, 3, 6, 8, 10, 1, 8, 512, false, true, true, TW_LDS, 2, false),
#elif defined(MIFFT_ALT_ROWS480) && MIFFT_ALT_ROWS480 == 2
    MIFFT_CFG_NTL("rows480_5x4x4x6_t8x512", float, MIFFT_F32, 480, 4, 5, 4, 4, 6, 8, 512, false, true, true, TW_LDS, 2, false),
#elif defined(MIFFT_ALT_ROWS480) && MIFFT_ALT_ROWS480 == 3
    MIFFT_CFG_NTL("rows480_10x6x8_t8x384", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 384, false, true, true, TW_LDS, 2, false),
#elif defined(MIFFT_ALT_ROWS480) && MIFFT_ALT_ROWS480 == 4
    MIFFT_CFG_NTL("rows480_4x4x5x6_t4x256", float, MIFFT_F32, 480, 4, 4, 4, 5, 6, 4, 256, false, true, true, TW_LDS, 2, false),
#else
    MIFFT_CFG_NTL("rows480_10x6x8", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
#endif
    MIFFT_CFG_NTL("rows640_10x8x8", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_NTL_R("rows480_10x6x8", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_MID_ST("rows480_10x6x8", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_MID_ST_R("rows480_10x6x8", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_CR("rows480_10x6x8", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_NTL_R("rows640_10x8x8", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_MID_ST("rows640_10x8x8", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_MID_ST_R("rows640_10x8x8", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_CR("rows640_10x8x8", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    // ---- strided dimensions, fp32 (in place, LDS column tiles) ----
    // ten waves, one per sub-problem of the radix-10 first pass (WSUB): the 8 x 8 passes exchange without workgroup
    // barriers, the waves drift apart and overlap each other's HBM traffic; with the next tile's loads issued in slices
    // between the passes 0.1009 -> 0.0969 ms for 100 x 640 x 480 (tools/tune GROUP 20)
    // Hermitian twins (last pass of a real-input 2-D .. 4-D plan; listed first, taken only when the pass asks for one)
    MIFFT_CFG_WSUB_HERM("cols640_10x8x8_ws", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_LDS, 1, true),
    MIFFT_CFG_WSUB_HERM("cols480_10x6x8_ws", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 16, 640, true, true, true, TW_LDS, 1, true),
    MIFFT_CFG_HERM("cols128_16x8_w32", float, MIFFT_F32, 128, 2, 16, 8, 1, 1, 32, 512, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG_HERM("cols128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG_HERM("cols64_8x8_w64", float, MIFFT_F32, 64, 2, 8, 8, 1, 1, 64, 512, true, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HERM("cols64_8x8_w32", float, MIFFT_F32, 64, 2, 8, 8, 1, 1, 32, 256, true, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HERM("cols64_4x4x4", float, MIFFT_F32, 64, 3, 4, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG_HERM("cols256_16x16", float, MIFFT_F32, 256, 2, 16, 16, 1, 1, 16, 256, true, true, true, TW_LDS, 2, false),
    // half-store twins (the pass before a Hermitian last pass stores only the lower half of its dimension; taken only
    // when the scheduler asks for one): the row pass of a real-input 2-D plan, the middle column pass of a 3-D / 4-D one
    MIFFT_CFG_HS_X(true, 1, 2, false, "rows480_10x6x8_r_ntl", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_HS_X(true, 0, -1, false, "rows480_10x6x8_r", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_HS_X(true, 1, 2, false, "rows640_10x8x8_r_ntl", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    MIFFT_CFG_HS_X(true, 0, -1, false, "rows640_10x8x8_r", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
    // ... power-of-two rows: the streaming twin (stream_pref 1) and the plain kernel
    MIFFT_CFG_HS_X(true, 3, 1, false, "rows1024_16x8x8_r_nt", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
    MIFFT_CFG_HS_X(true, 0, -1, false, "rows1024_16x8x8_r", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
    MIFFT_CFG_HS_X(true, 3, 1, false, "rows512_8x8x8_r_nt", float, MIFFT_F32, 512, 3, 8, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HS_X(true, 0, -1, false, "rows512_8x8x8_r", float, MIFFT_F32, 512, 3, 8, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HS_X(true, 3, 1, false, "rows256_8x8x4_r_nt", float, MIFFT_F32, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HS_X(true, 0, -1, false, "rows256_8x8x4_r", float, MIFFT_F32, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HS_X(true, 3, 1, false, "rows2048_16x16x8_r_nt", float, MIFFT_F32, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HS_X(true, 0, -1, false, "rows2048_16x16x8_r", float, MIFFT_F32, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HS("cols128_16x8_w32", float, MIFFT_F32, 128, 2, 16, 8, 1, 1, 32, 512, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG_HS("cols128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG_HS("cols64_8x8_w64", float, MIFFT_F32, 64, 2, 8, 8, 1, 1, 64, 512, true, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HS("cols64_8x8_w32", float, MIFFT_F32, 64, 2, 8, 8, 1, 1, 32, 256, true, true, true, TW_LDS, 2, false),
    MIFFT_CFG_HS("cols64_4x4x4", float, MIFFT_F32, 64, 3, 4, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG_HS("cols256_16x16", float, MIFFT_F32, 256, 2, 16, 16, 1, 1, 16, 256, true, true, true, TW_LDS, 2, false),
    MIFFT_CFG_WSUB("cols640_10x8x8_ws", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_LDS, 1, true),
    MIFFT_CFG_WSUB("cols480_10x6x8_ws", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 16, 640, true, true, true, TW_LDS, 1, true),
    // 16 columns x 1024 points = 128 KiB: the four-step passes of 2^20-point transforms (0.266 vs 0.349 ms for
    // the generated 8-column tile at 64 x 2^20)
    MIFFT_CFG("cols1024_16x8x8", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, false),
    // transposed + twiddled stores: the first pass of the two-pass four-step (N = N1 * N2, N1 one of these)
    MIFFT_CFG_TS("cols4096_8x8x8x8", float, MIFFT_F32, 4096, 4, 8, 8, 8, 8, 4, 512, true, true, false, TW_GLOBAL, 1, false),
    MIFFT_CFG_TS("cols2048_8x16x16", float, MIFFT_F32, 2048, 3, 8, 16, 16, 1, 8, 512, true, true, false, TW_LDS, 1, false),
    MIFFT_CFG_TS("cols1024_16x8x8", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 16, 512, true, true, false, TW_LDS, 1, false),
    MIFFT_CFG_TS("cols512_8x8x8", float, MIFFT_F32, 512, 3, 8, 8, 8, 1, 16, 512, true, true, false, TW_LDS, 1, false),
    // 32-column tiles for the four-step first pass (256-byte runs on the read side; whole tiles only): 3906 x 16384
    // 0.412 -> 0.398 ms, 976 x 65536 0.405 -> 0.401 ms; 64 columns lose (0.420)
    MIFFT_CFG_TS("cols256_16x16_w32", float, MIFFT_F32, 256, 2, 16, 16, 1, 1, 32, 512, true, true, false, TW_LDS, 2, false),
    MIFFT_CFG_TS("cols128_16x8_w32", float, MIFFT_F32, 128, 2, 16, 8, 1, 1, 32, 512, true, true, false, TW_LDS, 4, false),
    MIFFT_CFG_TS("cols256_16x16", float, MIFFT_F32, 256, 2, 16, 16, 1, 1, 16, 256, true, true, false, TW_LDS, 2, false),
    MIFFT_CFG_TS("cols128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, true, true, false, TW_LDS, 4, false),
    MIFFT_CFG_TS("cols64_4x4x4", float, MIFFT_F32, 64, 3, 4, 4, 4, 1, 16, 256, true, true, false, TW_LDS, 4, false),
    // 32 columns = 256-byte runs: the z axis of 10 x 128^3 0.0597 -> 0.0537 ms (tools/tune GROUP 18); taken when the
    // stride is a multiple of 32 columns, the 16-column tile below otherwise
    MIFFT_CFG("cols128_16x8_w32", float, MIFFT_F32, 128, 2, 16, 8, 1, 1, 32, 512, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG("cols128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG("cols64_8x8_w64", float, MIFFT_F32, 64, 2, 8, 8, 1, 1, 64, 512, true, true, true, TW_LDS, 2, false),
    MIFFT_CFG("cols64_8x8_w32", float, MIFFT_F32, 64, 2, 8, 8, 1, 1, 32, 256, true, true, true, TW_LDS, 2, false),
    MIFFT_CFG("cols64_4x4x4", float, MIFFT_F32, 64, 3, 4, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG("cols256_16x16", float, MIFFT_F32, 256, 2, 16, 16, 1, 1, 16, 256, true, true, true, TW_LDS, 2, false),
    // ---- fp64 (the reference's own tests run in float64, fft/tests.mojo:394-417): same template, 16-byte
    //      elements; smaller butterflies per pass keep the live registers under 128 ----
    // streaming twins (tools/tune GROUPs 30-32, 100k x 1024 / 500k x 128 / 500k x 93 fp64): 0.631 -> 0.597 ms,
    // 0.360 -> 0.338 ms, 0.423 -> 0.360 ms
    MIFFT_CFG_STREAM("rows1024_f64_16x8x8", double, MIFFT_F64, 1024, 3, 16, 8, 8, 1, 2, 128, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM("rows128_f64_8x16", double, MIFFT_F64, 128, 2, 8, 16, 1, 1, 16, 256, false, true, true, TW_LDS, 2, false),
    MIFFT_CFG_STREAM_ST("rows93_f64_31x3_t64", double, MIFFT_F64, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST("rows1024_f64_4x4x8x8", double, MIFFT_F64, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST_R("rows1024_f64_4x4x8x8", double, MIFFT_F64, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_CR("rows1024_f64_4x4x8x8", double, MIFFT_F64, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST("rows512_f64_8x8x8", double, MIFFT_F64, 512, 3, 8, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST_R("rows512_f64_8x8x8", double, MIFFT_F64, 512, 3, 8, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_CR("rows512_f64_8x8x8", double, MIFFT_F64, 512, 3, 8, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST("rows256_f64_4x8x8", double, MIFFT_F64, 256, 3, 4, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST_R("rows256_f64_4x8x8", double, MIFFT_F64, 256, 3, 4, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_CR("rows256_f64_4x8x8", double, MIFFT_F64, 256, 3, 4, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST("rows128_f64_8x4x4", double, MIFFT_F64, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST_R("rows128_f64_8x4x4", double, MIFFT_F64, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_CR("rows128_f64_8x4x4", double, MIFFT_F64, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST("rows64_f64_4x4x4", double, MIFFT_F64, 64, 3, 4, 4, 4, 1, 32, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST_R("rows64_f64_4x4x4", double, MIFFT_F64, 64, 3, 4, 4, 4, 1, 32, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_CR("rows64_f64_4x4x4", double, MIFFT_F64, 64, 3, 4, 4, 4, 1, 32, 256, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_SMALL_ST("rows93_f64_31x3", double, MIFFT_F64, 93, 2, 31, 3, 1, 1, 32, 96, false, true, false, TW_LDS, 1, false),
    MIFFT_CFG_SMALL_ST_R("rows93_f64_31x3", double, MIFFT_F64, 93, 2, 31, 3, 1, 1, 32, 96, false, true, false, TW_LDS, 1, false),
    MIFFT_CFG_CR("rows93_f64_31x3", double, MIFFT_F64, 93, 2, 31, 3, 1, 1, 32, 96, false, true, false, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST("rows480_f64_10x6x8", double, MIFFT_F64, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST_R("rows480_f64_10x6x8", double, MIFFT_F64, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_CR("rows480_f64_10x6x8", double, MIFFT_F64, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST("rows640_f64_10x8x8", double, MIFFT_F64, 640, 3, 10, 8, 8, 1, 4, 128, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_MID_ST_R("rows640_f64_10x8x8", double, MIFFT_F64, 640, 3, 10, 8, 8, 1, 4, 128, false, true, true, TW_LDS, 1, false),
    MIFFT_CFG_CR("rows640_f64_10x8x8", double, MIFFT_F64, 640, 3, 10, 8, 8, 1, 4, 128, false, true, true, TW_LDS, 1, false),
    // tools/tune GROUPs 33 / 35 (fp64 100 x 640 x 480, 10 x 128^3): wave-owned sub-problems + sliced prefetch 0.226 ->
    // 0.179 ms; 32-column tiles (512-byte runs) 0.135 -> 0.116 ms, 16-column tiles 0.123 ms
    MIFFT_CFG_WSUB("cols640_f64_10x8x8_ws", double, MIFFT_F64, 640, 3, 10, 8, 8, 1, 8, 640, true, true, true, TW_LDS, 1, true),
    MIFFT_CFG_WSUB("cols480_f64_10x6x8_ws", double, MIFFT_F64, 480, 3, 10, 6, 8, 1, 8, 640, true, true, true, TW_LDS, 1, true),
    MIFFT_CFG("cols128_f64_16x8_w32", double, MIFFT_F64, 128, 2, 16, 8, 1, 1, 32, 512, true, true, true, TW_LDS, 2, false),
    MIFFT_CFG("cols128_f64_16x8_t16", double, MIFFT_F64, 128, 2, 16, 8, 1, 1, 16, 256, true, true, true, TW_LDS, 4, false),
    MIFFT_CFG("cols640_f64_4x4x8x5", double, MIFFT_F64, 640, 4, 4, 4, 8, 5, 8, 256, true, true, true, TW_LDS, 1, false),
    MIFFT_CFG("cols480_f64_4x4x6x5", double, MIFFT_F64, 480, 4, 4, 4, 6, 5, 8, 256, true, true, true, TW_LDS, 1, false),
    MIFFT_CFG("cols128_f64_8x4x4", double, MIFFT_F64, 128, 3, 8, 4, 4, 1, 8, 128, true, true, true, TW_LDS, 1, false),
    MIFFT_CFG("cols64_f64_4x4x4", double, MIFFT_F64, 64, 3, 4, 4, 4, 1, 8, 128, true, true, true, TW_LDS, 1, false),
    MIFFT_CFG("cols256_f64_4x8x8", double, MIFFT_F64, 256, 3, 4, 8, 8, 1, 8, 256, true, true, true, TW_LDS, 1, false),
};

template <class CR, class CC>
static int launch_plane(const Plan& plan, const DimPass& pass, const void* in, void* out, int64_t count,
                        hipStream_t stream) {
    if (count == 0) return MIFFT_OK;
    TileParams tp{};
    tp.in = in;
    tp.out = out;
    tp.tw = pass.d_twiddle;
    tp.inverse = plan.inverse;
    tp.scale = plan.inverse ? 1.0 / ((double)pass.N * (double)pass.N1) : 1.0;
    tp.inner = CC::TILE;  // column stride inside a plane = N2
    tp.tiles_per_outer = 1;
    tp.n_rows = 0;
    tp.n_tiles = count * pass.outer;  // planes
    tp.reverse = pass.reverse;
    tp.store_lim = pass.store_lim;  // (HS column sides)
    auto k = plane_kernel<CR, CC>;
    const long long grid = tile_grid<CR>(plan.num_cus, tp.n_tiles, pass.wg_per_cu);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(CR::THREADS), CR::LDS_BYTES, stream, tp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_error(e, "plane_kernel launch");
    return MIFFT_OK;
}

// wave-private variant (plane_kernel_wp): LDS pitch N2 + PAD
template <class CR, class CC, int PAD>
static int launch_plane_wp(const Plan& plan, const DimPass& pass, const void* in, void* out, int64_t count,
                           hipStream_t stream) {
    if (count == 0) return MIFFT_OK;
    using G = WavePlane<CR, CC, PAD>;
    TileParams tp{};
    tp.in = in;
    tp.out = out;
    tp.tw = pass.d_twiddle;
    tp.inverse = plan.inverse;
    tp.scale = plan.inverse ? 1.0 / ((double)pass.N * (double)pass.N1) : 1.0;
    tp.inner = CC::TILE;
    tp.tiles_per_outer = 1;
    tp.n_tiles = count * pass.outer;  // planes
    tp.reverse = pass.reverse;
    tp.store_lim = pass.store_lim;  // (HS column sides)
    long long per_cu = (160 * 1024) / (long long)G::LDS_BYTES;
    if (per_cu > 2048 / CR::THREADS) per_cu = 2048 / CR::THREADS;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)plan.num_cus * per_cu;
    if (grid > tp.n_tiles) grid = tp.n_tiles;
    hipLaunchKernelGGL((plane_kernel_wp<CR, CC, PAD>), dim3((unsigned)grid), dim3(CR::THREADS), G::LDS_BYTES, stream, tp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_error(e, "plane_kernel_wp launch");
    return MIFFT_OK;
}

// plane_kernel_wp<.., FS = true>: rows of N1 * N2 points as a four-step inside one LDS plane
template <class CR, class CC, int PAD>
static int launch_row2d(const Plan& plan, const DimPass& pass, const void* in, void* out, int64_t count, hipStream_t stream) {
    if (count == 0) return MIFFT_OK;
    using G = WavePlane<CR, CC, PAD>;
    TileParams tp{};
    tp.in = in;
    tp.out = out;
    tp.tw = pass.d_twiddle;  // row side, W_N2
    tp.tlo = pass.d_aux;     // column side, W_N1 (rectangular planes)
    tp.thi = pass.d_aux3;    // W_M, forward
    tp.inverse = plan.inverse;
    tp.scale = plan.inverse ? 1.0 / (double)pass.N : 1.0;
    tp.inner = CC::TILE;
    tp.tiles_per_outer = 1;
    tp.n_tiles = count * pass.outer;  // rows
    tp.reverse = 0;
    constexpr size_t LDS = G::LDS_BYTES + (size_t)(G::N1 + G::N2) * 2 * sizeof(typename CR::T);
    long long grid = (long long)plan.num_cus * ((160 * 1024) / (long long)LDS > 0 ? (160 * 1024) / (long long)LDS : 1);
    if (grid > tp.n_tiles) grid = tp.n_tiles;
    hipLaunchKernelGGL((plane_kernel_wp<CR, CC, PAD, true>), dim3((unsigned)grid), dim3(CR::THREADS), LDS, stream, tp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_error(e, "plane_kernel_wp<FS> launch");
    return MIFFT_OK;
}

template <class CR, class CC, int PAD>
static int prepare_row2d() {
    using G = WavePlane<CR, CC, PAD>;
    constexpr size_t LDS = G::LDS_BYTES + (size_t)(G::N1 + G::N2) * 2 * sizeof(typename CR::T);
    hipError_t e = hipFuncSetAttribute((const void*)plane_kernel_wp<CR, CC, PAD, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    return MIFFT_OK;
}

template <class CR, class CC, int PAD>
static int prepare_plane_wp() {
    using G = WavePlane<CR, CC, PAD>;
    if (G::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)plane_kernel_wp<CR, CC, PAD>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    return MIFFT_OK;
}

template <class CR, class CC>
static int prepare_plane() {
    if (CR::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)plane_kernel<CR, CC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)CR::LDS_BYTES);
        if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    return MIFFT_OK;
}

struct PlaneEntry {
    bool ntl;      // non-temporal loads of x: first pass of a cache-resident N-D transform
    bool nts;      // non-temporal stores: the plane is the ONLY pass (2-D plan) and the batch sits in the window of 3.1c
    bool in_real;  // reads a real (C_in = 1) tensor
    int out_dtype;
    int N1, N2;
    const char* name;
    LaunchFn launch;
    int (*prepare)();
    int threads;
    size_t lds;
    bool hs = false;  // the column side stores only rows 0 .. N1 / 2 of the plane (TileCfg::HS): a Hermitian pass follows
};

// rows configuration (N2, TILE = N1, HBM -> LDS) then columns configuration (N1, TILE = N2, LDS -> HBM).
// A 128 x 128 plane = 128 KiB leaves ONE workgroup per CU: it pays only with the next plane prefetched into
// registers and with per-plane recomputed offsets (tile_kernel.h, OPAQUE_TID) -- before that change the same kernel
// spilled and took 0.121 ms for z+y of 10 x 128^3 against 0.060 + 0.061 ms for the two separate passes; now
// 0.080 ms (tools/tune/tune_tile.hip GROUP 7).
using Plane64R = TileCfg<float, 64, 2, 8, 8, 1, 1, 64, 512, false, true, false, TW_LDS, 2, false>;
using Plane64C = TileCfg<float, 64, 2, 8, 8, 1, 1, 64, 512, true, false, true, TW_LDS, 2, false>;
using Plane128R = TileCfg<float, 128, 2, 16, 8, 1, 1, 128, 1024, false, true, false, TW_LDS, 4, true>;
using Plane128C = TileCfg<float, 128, 2, 16, 8, 1, 1, 128, 1024, true, false, true, TW_LDS, 4, false>;
// wave-private variant: radix 8 first (pass 0 reads 128-byte runs), the next plane's loads sliced between the passes:
// 1280 planes 0.0688 ms against 0.0711 for 16 x 8 (tools/tune GROUP 7)
using Plane128WR = TileCfg<float, 128, 2, 8, 16, 1, 1, 128, 1024, false, true, false, TW_LDS, 4, true>;
using Plane128WC = TileCfg<float, 128, 2, 8, 16, 1, 1, 128, 1024, true, false, true, TW_LDS, 4, false>;
using Plane64RN = TileCfg<float, 64, 2, 8, 8, 1, 1, 64, 512, false, true, false, TW_LDS, 2, false, 0, false, false, 1>;
using Plane128WRN = TileCfg<float, 128, 2, 8, 16, 1, 1, 128, 1024, false, true, false, TW_LDS, 4, true, 0, false, false, 1>;
// column sides with non-temporal stores (a 2-D plan whose only pass is the plane)
using Plane64CS = TileCfg<float, 64, 2, 8, 8, 1, 1, 64, 512, true, false, true, TW_LDS, 2, false, 0, false, false, 2>;
using Plane128WCS = TileCfg<float, 128, 2, 8, 16, 1, 1, 128, 1024, true, false, true, TW_LDS, 4, false, 0, false, false, 2>;
// real-input twins (C_in = 1 promoted in the pass-0 load)
using Plane64RR = TileCfg<float, 64, 2, 8, 8, 1, 1, 64, 512, false, true, false, TW_LDS, 2, false, 0, true>;
using Plane128WRR = TileCfg<float, 128, 2, 8, 16, 1, 1, 128, 1024, false, true, false, TW_LDS, 4, true, 0, true>;
// half-store column sides (the plane of a real-input 3-D plan whose last pass is a Hermitian twin)
using Plane64CH = TileCfg<float, 64, 2, 8, 8, 1, 1, 64, 512, true, false, true, TW_LDS, 2, false, 0, false, false, 0, false, float, false, false, 0, false, true>;
using Plane128WCH = TileCfg<float, 128, 2, 8, 16, 1, 1, 128, 1024, true, false, true, TW_LDS, 4, false, 0, false, false, 0, false, float, false, false, 0, false, true>;

static const PlaneEntry kPlaneTable[] = {
    {true, false, false, MIFFT_F32, 64, 64, "plane64x64_8x8_ntl", launch_plane<Plane64RN, Plane64C>,
     prepare_plane<Plane64RN, Plane64C>, 512, Plane64RN::LDS_BYTES},
    {false, true, false, MIFFT_F32, 64, 64, "plane64x64_8x8_nts", launch_plane<Plane64R, Plane64CS>, prepare_plane<Plane64R, Plane64CS>,
     512, Plane64R::LDS_BYTES},
    {false, false, false, MIFFT_F32, 64, 64, "plane64x64_8x8", launch_plane<Plane64R, Plane64C>, prepare_plane<Plane64R, Plane64C>,
     512, Plane64R::LDS_BYTES},
    // real input is not bound by its bytes (DESIGN_EXPERIMENTS.md R3.8): the wave-private exchanges, a tie for complex 64 x 64
    // planes, win here -- 6400 real planes 0.0499 -> 0.0447 ms (tools/tune GROUP 28)
    {false, false, true, MIFFT_F32, 64, 64, "plane64x64_8x8_wp_r_hs", launch_plane_wp<Plane64RR, Plane64CH, 8>,
     prepare_plane_wp<Plane64RR, Plane64CH, 8>, 512, WavePlane<Plane64RR, Plane64CH, 8>::LDS_BYTES, true},
    {false, false, true, MIFFT_F32, 64, 64, "plane64x64_8x8_r_hs", launch_plane<Plane64RR, Plane64CH>, prepare_plane<Plane64RR, Plane64CH>,
     512, Plane64RR::LDS_BYTES, true},
    {false, false, true, MIFFT_F32, 64, 64, "plane64x64_8x8_r", launch_plane<Plane64RR, Plane64C>, prepare_plane<Plane64RR, Plane64C>,
     512, Plane64RR::LDS_BYTES},
    // wave-private exchanges (plane_kernel_wp): 2 workgroup barriers per plane instead of 12; 1280 planes 0.0812 ->
    // 0.0744 ms (tools/tune GROUP 7).  For 64 x 64 planes (four workgroups per CU already overlap) it ties.
    // (next-plane register prefetch already keeps its loads far ahead: the hint pays only from ~210 MB per tensor --
    //  168 / 185 / 235 MB: 0.1164 / 0.1296 / 0.1671 ms plain, 0.1177 / 0.1325 / 0.1588 ms with it)
    {true, false, false, MIFFT_F32, 128, 128, "plane128x128_8x16_wp_ntl", launch_plane_wp<Plane128WRN, Plane128WC, 8>,
     prepare_plane_wp<Plane128WRN, Plane128WC, 8>, 1024, WavePlane<Plane128WRN, Plane128WC, 8>::LDS_BYTES},
    {false, true, false, MIFFT_F32, 128, 128, "plane128x128_8x16_wp_nts", launch_plane_wp<Plane128WR, Plane128WCS, 8>,
     prepare_plane_wp<Plane128WR, Plane128WCS, 8>, 1024, WavePlane<Plane128WR, Plane128WCS, 8>::LDS_BYTES},
    {false, false, false, MIFFT_F32, 128, 128, "plane128x128_8x16_wp", launch_plane_wp<Plane128WR, Plane128WC, 8>,
     prepare_plane_wp<Plane128WR, Plane128WC, 8>, 1024, WavePlane<Plane128WR, Plane128WC, 8>::LDS_BYTES},
    {false, false, true, MIFFT_F32, 128, 128, "plane128x128_8x16_wp_r_hs", launch_plane_wp<Plane128WRR, Plane128WCH, 8>,
     prepare_plane_wp<Plane128WRR, Plane128WCH, 8>, 1024, WavePlane<Plane128WRR, Plane128WCH, 8>::LDS_BYTES, true},
    {false, false, true, MIFFT_F32, 128, 128, "plane128x128_8x16_wp_r", launch_plane_wp<Plane128WRR, Plane128WC, 8>,
     prepare_plane_wp<Plane128WRR, Plane128WC, 8>, 1024, WavePlane<Plane128WRR, Plane128WC, 8>::LDS_BYTES},
};

bool select_fast_plane(const Plan& plan, DimPass& pass) {
    if (plan.in_dtype != plan.out_dtype) return false;
    for (const PlaneEntry& e : kPlaneTable) {
        if (e.out_dtype != plan.out_dtype || e.N2 != pass.N || e.N1 != pass.N1) continue;
        if (e.in_real != (pass.first && plan.in_components == 1)) continue;
        if (e.hs != pass.want_half) continue;
        if (e.ntl && !(plan.cache_resident_nd && plan.ndim > 2)) continue;  // a 2-D plane is the only pass: nothing to keep
        if (e.ntl && e.N1 == 128 &&
            plan.size_batch() * (double)plan.prod * (double)plan.out_elem_bytes() < config().nd_plane128_min_bytes)
            continue;
        if (e.nts && !(plan.ndim == 2 && nts_window_bytes(plan.size_batch() * (double)plan.prod * (double)plan.out_elem_bytes() * 2.0)))
            continue;
        pass.kernel_name = e.name;
        pass.launch = e.launch;
        pass.prepare = e.prepare;
        pass.tile = 1;
        pass.threads = e.threads;
        pass.lds_bytes = e.lds;
        pass.ld = (int)pass.N;
        pass.hs = e.hs;
        return true;
    }
    return false;
}

// Long contiguous dimensions as a four-step INSIDE one LDS plane (plane_kernel_wp<.., FS>): 16384 = 128 x 128 and 8192 = 64
// rows x 128 columns, complex fp32 in and out.  One launch, twiddles of the two short sides + a two-level table of N1 + N2
// entries in LDS, against one workgroup per row reading a length-M table in every pass (16384: global table, 2 TB/s;
// 8192: 64 KB of LDS for the table, one workgroup per CU, 4 TB/s) or two column-tile launches for big batches (2.4 TB/s
// effective).  3906 x 16384 0.385 -> 0.246 ms, 100 x 16384 0.0241 -> 0.0167 ms.  MIFFT_ROW2D=0 turns it off.
using Row2D64R = TileCfg<float, 128, 2, 8, 16, 1, 1, 64, 512, false, true, false, TW_LDS, 2, true>;
using Row2D64C = TileCfg<float, 64, 2, 8, 8, 1, 1, 128, 512, true, false, true, TW_LDS, 2, false>;

using Row2D64RD = TileCfg<double, 128, 2, 8, 16, 1, 1, 64, 512, false, true, false, TW_LDS, 1, true>;
using Row2D64CD = TileCfg<double, 64, 2, 8, 8, 1, 1, 128, 512, true, false, true, TW_LDS, 1, false>;

struct Row2DEntry {
    int dtype;
    bool in_real;
    int64_t M;
    int N1;  // rows of the plane = column-side length
    const char* name;
    LaunchFn launch;
    int (*prepare)();
    int threads;
    size_t lds;
};
static const Row2DEntry kRow2DTable[] = {
    {MIFFT_F32, false, 16384, 128, "rows16384_fs128x128_wp", launch_row2d<Plane128WR, Plane128WC, 8>, prepare_row2d<Plane128WR, Plane128WC, 8>,
     1024, WavePlane<Plane128WR, Plane128WC, 8>::LDS_BYTES + 256 * 8},
    {MIFFT_F32, false, 8192, 64, "rows8192_fs64x128_wp", launch_row2d<Row2D64R, Row2D64C, 8>, prepare_row2d<Row2D64R, Row2D64C, 8>, 512,
     WavePlane<Row2D64R, Row2D64C, 8>::LDS_BYTES + 192 * 8},
    // (4096 = 64 x 64 measured too: 4.8 TB/s at 50k rows against 5.4 for the four-pass row kernel -- not taken)
    {MIFFT_F64, false, 8192, 64, "rows8192_f64_fs64x128_wp", launch_row2d<Row2D64RD, Row2D64CD, 4>, prepare_row2d<Row2D64RD, Row2D64CD, 4>,
     512, WavePlane<Row2D64RD, Row2D64CD, 4>::LDS_BYTES + 192 * 16},
    {MIFFT_F32, true, 16384, 128, "rows16384_fs128x128_wp_r", launch_row2d<Plane128WRR, Plane128WC, 8>, prepare_row2d<Plane128WRR, Plane128WC, 8>,
     1024, WavePlane<Plane128WRR, Plane128WC, 8>::LDS_BYTES + 256 * 8},
    // (real input at 8192 points loses 10-19 % against the runtime-specialised row kernel: 32-byte runs on the load side)
};

bool select_row2d(const Plan& plan, DimPass& pass) {
    if (!config().row2d) return false;  // (lab switch; always on in the product library)
    if (pass.inner != 1 || !pass.first || plan.in_dtype != plan.out_dtype) return false;
    for (const Row2DEntry& e : kRow2DTable) {
        if (e.M != pass.N || e.dtype != plan.out_dtype || e.in_real != (plan.in_components == 1)) continue;
        pass.kernel_name = e.name;
        pass.launch = e.launch;
        pass.prepare = e.prepare;
        pass.tile = 1;
        pass.threads = e.threads;
        pass.lds_bytes = e.lds;
        pass.ld = (int)pass.N;
        pass.N1 = e.N1;
        pass.row2d_m = e.M;
        pass.plane_needs_tw1 = (int64_t)e.N1 * e.N1 != e.M;  // rectangular: the column side has its own table (d_aux)
        return true;
    }
    return false;
}

bool select_fast_tstore(const Plan& plan, DimPass& pass) {
    for (const FastEntry& e : kFastTable) {
        if (!e.tstore || e.out_dtype != plan.out_dtype || e.N != pass.N || pass.inner < e.tile) continue;
        if (e.tile > 16 && pass.inner % e.tile != 0) continue;  // wide tiles: whole tiles only
        pass.kernel_name = e.name;
        pass.launch = e.launch;
        pass.prepare = e.prepare;
        pass.tile = e.tile;
        pass.threads = e.threads;
        pass.lds_bytes = e.lds;
        pass.ld = (int)pass.N;
        return true;
    }
    return false;
}

bool nts_window_bytes(double total_bytes) {
    return total_bytes > config().nts_min_bytes && total_bytes <= config().nts_max_bytes;
}
bool nts_window(const Plan& plan, double total_bytes) { return plan.ndim == 1 && nts_window_bytes(total_bytes); }

// Measured workgroups-per-CU of the persistent grid where the LDS / wave-count formula of tile_grid<> is not the best choice
// (tools/grid_sweep.py, one box, interleaved).  Name prefixes: a value applies to every twin of the configuration.
struct GridPerCu {
    const char* prefix;
    int per_cu;
};
static const GridPerCu kGridPerCu[] = {
    {"rows480_10x6x8", 3},   // 136-148 VGPRs: three workgroups resident, four launched left a quarter of the tiles to a thin tail
};

static int grid_per_cu_of(const char* name) {
    if (config().grid_per_cu > 0) return config().grid_per_cu;  // lab knob (tools/grid_sweep.py); 0 in the product library
    for (const GridPerCu& g : kGridPerCu)
        if (strncmp(name, g.prefix, strlen(g.prefix)) == 0) return g.per_cu;
    return 0;
}

bool select_fast(const Plan& plan, DimPass& pass) {
    // fast families read real or complex input of the output dtype; integer input and mixed
    // precision run on the generic family
    if (pass.first && plan.in_dtype != plan.out_dtype) return false;
    const bool cols = pass.inner != 1;
    // read + write volume of one exec far beyond the 256-MB Infinity Cache -> non-temporal twins apply
    const double total_bytes = plan.size_batch() * (double)plan.prod * (double)plan.out_elem_bytes() * 2.0;
    const bool streaming = total_bytes > config().streaming_min_bytes;
    bool hand_table = true;  // the hand-tuned lengths keep their `_nts` twins up to the streaming threshold (0.6 GB), where
                             // the `_nt` twins take over: 50k x 1024 still gains 4-6 % with non-temporal stores
    auto try_entry = [&](const FastEntry& e) {
        if (e.out_dtype != plan.out_dtype || e.N != pass.N || e.cols != cols || e.tstore) return false;
        if (e.herm && !pass.want_herm) return false;  // Hermitian twins: only where the scheduler asks for one
        if (e.hs != pass.want_half) return false;       // half-store twins likewise, and nothing else when it does
        if (!e.herm && pass.herm_only) return false;
        if (e.herm && !herm_pays(plan, pass, e.tile, e.lds, e.threads)) return false;
        if (e.in_real != (pass.first && plan.in_components == 1)) return false;
        if (e.stream_pref == 1 && !streaming) return false;
        if (e.stream_pref == 2 && !(plan.cache_resident_nd && pass.first)) return false;
        if (e.stream_pref == 3 && !(nts_window(plan, total_bytes) ||
                                    (hand_table && plan.ndim == 1 && total_bytes > config().nts_min_bytes && !streaming)))
            return false;
        if (e.stream_pref == 4 && !(plan.ndim == 1 && total_bytes > config().nts_small_min_bytes)) return false;
        if (cols && e.tile > 16 && pass.inner % e.tile != 0) return false;  // wide tiles: whole tiles only
        if (cols && e.tile > 16 && pass.col_prefix % e.tile != 0) return false;  // (also of a transformed column prefix)
        // (a strided dimension with fewer columns than one tile still runs here: the ragged tile clamps its loads and
        //  masks its stores; the literal-stage alternative is an order of magnitude slower)
        pass.kernel_name = e.name;
        pass.launch = e.launch;
        pass.prepare = e.prepare;
        pass.tile = e.tile;
        pass.threads = e.threads;
        pass.lds_bytes = e.lds;
        pass.ld = (int)pass.N;
        pass.wg_per_cu = grid_per_cu_of(e.name);
        pass.hs = e.hs;
        pass.regime_twin = e.stream_pref > 0;
        pass.prefix_ok = e.cols && !e.tstore && !e.herm;
        pass.herm_d0 = pass.herm_d1 = pass.herm_d2 = 0;
        if (e.herm) herm_set_dims(plan, pass);  // trailing dimensions of the column space
        return true;
    };
    for (const FastEntry& e : kFastTable)  // hand-tuned entries win
        if (try_entry(e)) return true;
    hand_table = false;
    if (config().skip_gen_table) return false;  // (lab switch)
    int ngen = 0;
    const FastEntry* gen;
    if (plan.out_dtype == MIFFT_F64)
        gen = cols ? gen_cols_f64_table(&ngen) : gen_rows_f64_table(&ngen);
    else
        gen = cols ? gen_cols_table(&ngen) : gen_rows_table(&ngen);
    for (int i = 0; i < ngen; ++i)
        if (try_entry(gen[i])) return true;
    return false;
}

}  // namespace mifft
