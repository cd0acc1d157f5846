/* C ABI of libwavelets_hip.so - the MI355X (gfx950) separable 2-D wavelet filterbank engine.
 *
 * This is the drop-in boundary of DESIGN.md section (b).  The reference
 * (fbcotter/pytorch_wavelets v1.3.0) has no FFI: its per-scale operators are the
 * torch.autograd.Function classes named next to each entry point below, and their bodies are
 * ATen calls.  A maintainer of the reference replaces each Function's forward/backward body with
 * one call into this library (ctypes stub in INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers and sizes only; every data/tap pointer is a DEVICE pointer (HBM), tensors
 *     are dense NCHW with N and C collapsed into `planes`; nothing is allocated or freed here.
 *   - `dtype`: WL_F32 0, WL_F16 1 (fp32 accumulate), WL_F64 2.  Taps are float for F32/F16 data
 *     and double for F64 data, in the order the reference's module buffers hold them
 *     (analysis/DTCWT taps reversed, synthesis taps unreversed).
 *   - `mode`: the reference's integer codes (dwt/lowlevel.py:274-290): 0 zero, 1 symmetric,
 *     2 periodization, 4 reflect, 6 periodic.
 *   - `stream`: a hipStream_t (NULL = default stream).  Launches are asynchronous; no host sync.
 *   - return value: 0 on success, WL_ERR_* (<0) for argument errors, a positive hipError_t if the
 *     launch failed.  No exceptions cross the ABI, no global mutable state: re-entrant.
 */
#ifndef WAVELETS_HIP_H
#define WAVELETS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WL_F32 0
#define WL_F16 1
#define WL_F64 2

#define WL_ERR_MODE (-1)        /* unknown / unsupported padding mode                         */
#define WL_ERR_SHAPE (-2)       /* inconsistent sizes                                         */
#define WL_ERR_UNSUPPORTED (-3) /* valid for the reference but not implemented by this engine */
#define WL_ERR_DTYPE (-4)
#define WL_ERR_TAPS (-5)        /* tap count out of range (1..WL_MAX_TAPS)                    */
#define WL_MAX_TAPS 128

/* library version (major*10000 + minor*100 + patch) and build flavour ("hip-gfx950" / "emu").
 * ABI history: an entry point never changes its argument list under the same name.
 *   100, 200 (rounds 1-4): wl_dwt2d_analysis_fused / _synthesis_fused / _analysis_stream / _synthesis_stream WITHOUT device scratch.
 *   200 (round 5): those four names took a `tap_scratch` pointer in front of `stream` without a version change (a mistake: same
 *          symbol, other arguments).
 *   210 (this header): the four names are back to their round-4 argument lists; the lattice variants that need device scratch are
 *          reached through the new *_ex entry points (tap_scratch + the caller-owned `tap_state`), and
 *          wl_dwt2d_analysis_fused_strided is gone (wl_dwt2d_analysis_fused_ex takes the strides).  A caller built against 2.0.0
 *          must be recompiled against this header (check wl_version() >= 210). */
int wl_version(void);
const char* wl_backend(void);

/* Diagnostics.  wl_set_option("generic_only", 1) routes every operator to the runtime-L generic kernels (the test-suite
 * compares the two kernel families); the initial value is read once from $WL_GENERIC_ONLY.  ("no_stream", 1) keeps the
 * DTCWT / ScatLayer entry points off their streaming kernels (tile kernels instead; wl_dtcwt_fwd_level12 then declines);
 * ("scat_stream", 1) lets wl_scat_fwd_level1 try the strip kernel (measured no faster: off by default).  Returns 0, or
 * WL_ERR_UNSUPPORTED for an unknown name.  wl_last_kernel(): name of the kernel functor launched last by any thread of
 * the process (static storage; autograd runs backward passes on its own threads), so that a benchmark can label its numbers with the dispatch actually taken. */
int wl_set_option(const char* name, int value);
int wl_get_option(const char* name);   /* current value, or WL_ERR_UNSUPPORTED for an unknown name */
const char* wl_last_kernel(void);
/* wl_launch_count(): kernels launched by the process so far; wl_kernel_history(back): the functor launched `back` launches
 * ago (0 = the last one, up to 31; "" beyond) - a multi-launch transform (DTCWT: one kernel per level or pair of levels)
 * can then be labelled launch by launch.  A name that contains "wl_launch_armed" is the two-bank variant queued behind a kernel
 * variant that relies on a relation between the filter banks (it returns at once unless the device finds the relation broken),
 * "wl_launch_aux" a helper launch (the one-thread examination of the banks in front of a lattice launch); wl_last_kernel names
 * neither.  wl_last_grid(): workgroups of the launch wl_last_kernel names (a test can see how a launcher cut the work: planes per
 * workgroup, row segments). */
long long wl_launch_count(void);
long long wl_last_grid(void);
const char* wl_kernel_history(int back);

/* Coefficient count of one 1-D analysis level: (n+L-1)/2, or (n+1)/2 for periodization.
 * Replaces pywt.dwt_coeff_len at dwt/lowlevel.py:153. */
int wl_dwt_coeff_len(int n, int L, int mode);

/* One 2-D analysis level = AFB2D.forward (dwt/lowlevel.py:336-347: afb1d along W, afb1d along H,
 * reshape + two .contiguous() copies), also SFB2D.backward (dwt/lowlevel.py:683-694).
 *   x (planes,H,W) -> ll (planes,Kh,Kw), highs (planes,3,Kh,Kw) with band order
 *   [W-lo/H-hi, W-hi/H-lo, W-hi/H-hi];  Kh = wl_dwt_coeff_len(H,Lh,mode), Kw likewise.
 *   h_w_*: the pair applied along W (first pair of AFB2D.forward), h_h_*: the pair along H. */
int wl_dwt2d_analysis(const void* x, void* ll, void* highs, int dtype, int64_t planes, int H, int W,
                      const void* h_w_lo, const void* h_w_hi, int Lw,
                      const void* h_h_lo, const void* h_h_hi, int Lh, int mode, void* stream);

/* The same with explicit layouts: x and ll as (planes, rows, cols) views with unit column stride and the given
 * plane / row strides (elements).  The level loop of DWTForward keeps the inner LL_j at a cache-line-aligned row
 * pitch (aligned stores for this level, aligned 8-byte loads for the next); highs and the final yl stay dense. */
int wl_dwt2d_analysis_strided(const void* x, int64_t x_plane_stride, int x_row_stride, void* ll,
                              int64_t ll_plane_stride, int ll_row_stride, void* highs, int dtype, int64_t planes,
                              int H, int W, const void* h_w_lo, const void* h_w_hi, int Lw, const void* h_h_lo,
                              const void* h_h_hi, int Lh, int mode, void* stream);

/* One 2-D synthesis level = SFB2D.forward (dwt/lowlevel.py:671-680: three sfb1d = six
 * conv_transpose2d + three adds), also AFB2D.backward (dwt/lowlevel.py:350-365, the crop is done
 * by passing a smaller OH/OW).
 *   ll (planes,Kh,Kw) read through explicit strides (elements) so that the reference's
 *   "unpad" slice ll[..., :-1, :] (dwt/transform2d.py:142-145) needs no copy;
 *   highs (planes,3,Kh,Kw) dense or NULL (= zeros, dwt/transform2d.py:137-139);
 *   y (planes,OH,OW) with OH <= 2*Kh-Lh+2 (2*Kh for periodization), OW likewise. */
int wl_dwt2d_synthesis(const void* ll, int64_t ll_plane_stride, int ll_row_stride, const void* highs,
                       void* y, int dtype, int64_t planes, int Kh, int Kw, int OH, int OW,
                       const void* g_w_lo, const void* g_w_hi, int Lw,
                       const void* g_h_lo, const void* g_h_hi, int Lh, int mode, void* stream);

/* `nlev` (1..3) analysis levels in ONE launch = the body of DWTForward.forward's level loop
 * (dwt/transform2d.py:63-74): x (planes,H,W) dense -> yh[j] (planes,3,H_j,W_j) for j < nlev and the last
 * level's low-pass yl; one workgroup streams one plane top to bottom, the intermediate LL_j stay in LDS rings and
 * HBM traffic is the algorithmic minimum.  `yh` is a HOST array of nlev device pointers.  Same taps (even length
 * L <= 12; 14 - 20 in the lattice form, see `strips`) on both axes of every level, F32/F16 data, float taps; rows of 16-byte multiples of up to
 * 2 KiB (F32: 3 KiB) and ~630 outputs;
 * zero / symmetric / reflect and (library version 210) periodization for nlev > 1 - even sizes below level 1, every level at least as long
 * as the filter -, any mode for nlev == 1.  `strips`: 0 = let the engine decide (it declines
 * below about 3/8 as many planes as compute units, where the tile kernels win; with fewer planes than compute units it
 * cuts planes in two so that every workgroup has a compute unit of its own), 1 = force this kernel, whole planes,
 * 2 = force, every plane cut in two; + 4 (bit 2) = a HINT that (h_h_lo, h_h_hi) hold the same taps as (h_w_lo, h_w_hi): the
 * 10- and 12-tap kernels then run their one-bank variant (one set of tap pairs in scalar registers), which compares the two
 * banks on the device first, with the two-bank variant queued behind it for the case that they differ (identical POINTERS for
 * both axes need no hint and no check); + 8 (bit 3) = a HINT that each highpass bank is the quadrature mirror of its lowpass bank.
 * The LATTICE variant of 10 - 20 taps (the only fused multi-level form of 14 - 20 taps) needs device scratch: see
 * wl_dwt2d_analysis_fused_ex; here those lengths return WL_ERR_UNSUPPORTED.  Returns WL_ERR_UNSUPPORTED outside the
 * kernel's envelope: the caller then uses wl_dwt2d_analysis level by level. */
int wl_dwt2d_analysis_fused(const void* x, void* yl, void* const* yh, int dtype, int64_t planes, int H,
                            int W, int nlev, const void* h_w_lo, const void* h_w_hi, const void* h_h_lo,
                            const void* h_h_hi, int L, int mode, int strips, void* stream);
/* The same with (a) x (planes, H, W) through a plane stride and a row pitch (elements; both whole 16-byte pieces, x 16-byte aligned):
 * the input of the level loop's SECOND and later iterations when the first level ran on another kernel (reference
 * dwt/transform2d.py:63-74: `ll` of one AFB2D.forward feeds the next).  W itself need not be a whole number of pieces then - the
 * odd-width ll below a 1024-wide image, 515 columns - as long as the row's last piece lies inside the pitch (the caller owns the
 * elements behind the row; the kernel overwrites what it loaded from there before any lane reads it);
 * and (b) device scratch for the LATTICE variant (csrc/wl_lattice.h): with BOTH hints (strips | 4 | 8) and tap_scratch =
 * WL_TAP_SCRATCH_BYTES of device memory that stay valid until the launches of this call have run (stream order), the 10- to
 * 20-tap kernels (10 - 20) run their lattice variant behind a one-thread examination of the banks (WlTapPrep), the
 * two-bank variant armed behind it.  tap_state: NULL, or a HOST int the caller keeps next to the scratch block - 0 when the
 * block is fresh, when any of the four tap pointers, their contents, L, the data type or the direction (analysis / synthesis)
 * changed; the library sets bit 0 when it has run the examination of exactly these banks into the block, bit 1 when that
 * examination included "one bank for both axes", and skips the one-thread kernel when the bits it needs are set (the levels of
 * one transform share one examination).  Only a launcher that actually ran the examination sets bits: a launch that has no use
 * for the scratch leaves *tap_state alone, so no later launch trusts an unexamined block. */
int wl_dwt2d_analysis_fused_ex(const void* x, int64_t x_plane_stride, int x_row_stride, void* yl, void* const* yh,
                               int dtype, int64_t planes, int H, int W, int nlev, const void* h_w_lo,
                               const void* h_w_hi, const void* h_h_lo, const void* h_h_hi, int L, int mode,
                               int strips, void* tap_scratch, int* tap_state, void* stream);

/* All `nlev` (1..3) synthesis levels in ONE launch = the body of DWTInverse.forward's level loop
 * (dwt/transform2d.py:131-148) = nlev x SFB2D.forward (dwt/lowlevel.py:671-680): yl (planes, Kh[nlev-1],
 * Kw[nlev-1]) with dense rows, planes `yl_plane_stride` elements apart (yl_h / yl_w = its logical size, which must
 * equal the coarsest high-pass size: the caller crops the reference's 'unpad' surplus row / column first);
 * yh[j] (planes,3,Kh[j],Kw[j]) dense, j = 0 finest (`yh`, `Kh`, `Kw` are HOST arrays of nlev entries) ->
 * y (planes, 2*Kh[0]-L+2, 2*Kw[0]-L+2) dense.  Between levels the low-pass of level j+1 is cropped to yh[j]'s size
 * exactly as the reference does.  One workgroup streams one plane (or half of one) coarsest level first; the
 * intermediate low-passes stay in LDS rings, every coefficient is read from HBM once (LDS-DMA in whole 1024-byte
 * chunks of the contiguous band planes) and x is written once.  Same even tap count L <= 12 on both axes of every
 * level, F32/F16 data, float taps; zero / symmetric / reflect / periodic, and (library version 210) periodization - there y is (planes, 2*Kh[0], 2*Kw[0]), every level exactly
 * twice the level above, yl the size of yh[nlev-1], at least L/2 coefficients per row and column (csrc/wl_idwt_rows.h, PER); row bytes and plane
 * bytes multiples of four.  `strips` (incl. the hint bits 2-3) as for wl_dwt2d_analysis_fused.  Returns WL_ERR_UNSUPPORTED
 * outside the kernel's envelope: the caller then uses wl_dwt2d_synthesis level by level. */
int wl_dwt2d_synthesis_fused(const void* yl, int64_t yl_plane_stride, int yl_row_stride, int yl_h, int yl_w,
                             const void* const* yh, const int* Kh, const int* Kw, void* y, int dtype,
                             int64_t planes, int nlev, const void* g_w_lo, const void* g_w_hi,
                             const void* g_h_lo, const void* g_h_hi, int L, int mode, int strips, void* stream);
/* The same with tap_scratch / tap_state as for wl_dwt2d_analysis_fused_ex: with both hints and the scratch the 8-20 tap kernels
 * run their lattice variant (the transposed recurrence of csrc/wl_lattice.h) - the only fused form of 14 - 20 taps. */
int wl_dwt2d_synthesis_fused_ex(const void* yl, int64_t yl_plane_stride, int yl_row_stride, int yl_h, int yl_w,
                                const void* const* yh, const int* Kh, const int* Kw, void* y, int dtype,
                                int64_t planes, int nlev, const void* g_w_lo, const void* g_w_hi,
                                const void* g_h_lo, const void* g_h_hi, int L, int mode, int strips, void* tap_scratch,
                                int* tap_state, void* stream);

/* Non-separable one-level analysis / synthesis with four Ly x Lx point-spread functions = afb2d_nonsep
 * (dwt/lowlevel.py:524-597) / sfb2d_nonsep (:746-798).  `f` / `g`: (4, Ly, Lx) device taps in the accumulate dtype,
 * exactly what prep_filt_afb2d_nonsep (already mirrored) / prep_filt_sfb2d_nonsep return.  Analysis: x (planes,H,W)
 * -> y (planes,4,Kh,Kw), K = wl_dwt_coeff_len; modes zero / symmetric / reflect / periodization (WL_ERR_MODE for
 * 'periodic', as upstream).  Synthesis: coeffs (planes,4,Kh,Kw) -> y (planes,OH,OW), OH/OW <= the natural size (a
 * crop).  Direct evaluation, one thread per output sample: completes the function-level API, not a tuned path. */
int wl_dwt2d_analysis_nonsep(const void* x, void* y, int dtype, int64_t planes, int H, int W, const void* f,
                             int Ly, int Lx, int mode, void* stream);
int wl_dwt2d_synthesis_nonsep(const void* coeffs, void* y, int dtype, int64_t planes, int Kh, int Kw, int OH,
                              int OW, const void* g, int Ly, int Lx, int mode, void* stream);

/* ---- DTCWT (filters = the reference's stored buffers: reversed columns, dtcwt/lowlevel.py:58-67) ------------ */

/* Level-1 forward = FWD_J1.forward -> fwd_j1 (dtcwt/transform_funcs.py:98-121, :346-358): 2 rowfilter +
 * 4 colfilter + 3 q2c + stacks.  x (planes,H,W); odd H/W are edge-replicated to He=H+(H&1), We likewise
 * (dtcwt/transform2d.py:116-120) inside the kernel.  ll (planes,He,We) or NULL; highs
 * (planes,6,He/2,We/2,2) in the reference's default o_dim=2/ri_dim=-1 layout, or NULL (skip_hps).
 * h0 (L0 taps) / h1 (L1 taps): odd lengths.  mode 1 = symmetric extension, any other valid code = zero padding
 * (dtcwt/lowlevel.py:75-79). */
int wl_dtcwt_fwd_level1(const void* x, void* ll, void* highs, int dtype, int64_t planes, int H, int W,
                        const void* h0, int L0, const void* h1, int L1, int mode, void* stream);

/* Level>=2 forward = FWD_J2PLUS.forward -> fwd_j2plus (transform_funcs.py:226-249, :380-392): 2 rowdfilt +
 * 4 coldfilt + q2c, always symmetric.  x (planes,H,W) with H,W even; if not multiples of 4 one row/col is
 * replicated on both sides (transform2d.py:131-135) inside the kernel: He=H+2*(H%4!=0).  ll (planes,He/2,We/2),
 * highs (planes,6,He/4,We/4,2) or NULL.  Four filters of even length L. */
int wl_dtcwt_fwd_level2(const void* x, void* ll, void* highs, int dtype, int64_t planes, int H, int W,
                        const void* h0a, const void* h0b, const void* h1a, const void* h1b, int L, void* stream);

/* Levels 1 AND 2 of the forward in ONE launch (csrc/wl_dtcwt_fused.h) = FWD_J1.forward followed by FWD_J2PLUS.forward the
 * way DTCWTForward.forward chains them (dtcwt/transform2d.py:121-141), the level-1 lowpass staying on chip: x (planes,H,W)
 * -> highs1 (planes,6,H/2,W/2,2), ll2 (planes,H/2,W/2), highs2 (planes,6,H/4,W/4,2).  Symmetric mode (1), H and W
 * multiples of 4, float32 / float16, the (5,7) and (5,3) level-1 pairs with 10-tap level-2 filters; the level-1 LOWPASS
 * FILTER MUST BE SYMMETRIC (h0o[t] == h0o[L0-1-t], true of every biorthogonal table of the reference: the rows above /
 * below the plane that level 2 reads are computed from the extended input instead of mirrored).  policy 0 = the engine
 * decides whether the launch pays, 1 = force.  Returns WL_ERR_UNSUPPORTED outside its envelope: callers then chain
 * wl_dtcwt_fwd_level1 and wl_dtcwt_fwd_level2. */
int wl_dtcwt_fwd_level12(const void* x, void* highs1, void* ll2, void* highs2, int dtype, int64_t planes, int H, int W,
                         const void* h0o, int L0, const void* h1o, int L1, const void* h0a, const void* h0b,
                         const void* h1a, const void* h1b, int LQ, int mode, int policy, void* stream);

/* Levels 2 AND 1 of the inverse in ONE launch (csrc/wl_dtcwt_inv_fused.h) = INV_J2PLUS.forward followed by INV_J1.forward the
 * way DTCWTInverse.forward chains its last two levels (dtcwt/transform2d.py:240-254 -> transform_funcs.py:279-307, :152-184),
 * the level-1 lowpass (H x W, the level-2 reconstruction) staying on chip: ll2 (planes,H/2,W/2) through strides, highs2
 * (planes,6,H/4,W/4,2), highs1 (planes,6,H/2,W/2,2) -> y (planes,H,W).  All three inputs present, symmetric mode (1), H and W
 * multiples of 4 (no 1-px crop between the levels), float32 / float16, the (7,5) / (5,7) level-1 pairs with 10-tap level-2
 * filters; any tap VALUES (nothing is assumed about them).  policy 0 = the engine decides whether the launch pays, 1 = force.
 * Returns WL_ERR_UNSUPPORTED outside its envelope: callers then chain wl_dtcwt_inv_level2 and wl_dtcwt_inv_level1. */
int wl_dtcwt_inv_level21(const void* ll2, int64_t ll2_plane_stride, int ll2_row_stride, const void* highs2,
                         const void* highs1, void* y, int dtype, int64_t planes, int H, int W, const void* g0o, int L0,
                         const void* g1o, int L1, const void* g0a, const void* g0b, const void* g1a, const void* g1b, int LQ,
                         int mode, int policy, void* stream);

/* Level-1 inverse = INV_J1.forward -> inv_j1 (transform_funcs.py:152-184, :419-431): c2q x 3, 4 colfilter,
 * 2 rowfilter, 3 adds.  ll (planes,H,W) through strides (so the 1-px crop of transform_funcs.py:171-176 is a
 * view) or NULL; highs (planes,6,H/2,W/2,2) or NULL; y (planes,H,W). */
int wl_dtcwt_inv_level1(const void* ll, int64_t ll_plane_stride, int ll_row_stride, const void* highs, void* y,
                        int dtype, int64_t planes, int H, int W, const void* g0, int L0, const void* g1, int L1,
                        int mode, void* stream);

/* Level>=2 inverse = INV_J2PLUS.forward -> inv_j2plus (transform_funcs.py:279-307, :455-468): c2q x 3,
 * 4 colifilt, 2 rowifilt, 3 adds.  ll (planes,h,w) through strides or NULL; highs (planes,6,h/2,w/2,2) or NULL;
 * y (planes,2h,2w). */
int wl_dtcwt_inv_level2(const void* ll, int64_t ll_plane_stride, int ll_row_stride, const void* highs, void* y,
                        int dtype, int64_t planes, int h, int w, const void* g0a, const void* g0b, const void* g1a,
                        const void* g1b, int L, void* stream);

/* ScatLayer forward = ScatLayerj1_f.forward (scatternet/lowlevel.py:76-111): level-1 DTCWT + 2x2 average of LL +
 * smoothed magnitude sqrt(re^2+im^2+b^2)-b, written as z (N,7,C,He/2,We/2) [(N,3+6,..) with combine_colour].
 * drdx/drdy (N,6,C,He/2,We/2) receive re/r, im/r for the backward pass, or NULL.  ll (N,C,He,We) or NULL: the
 * full-resolution level-1 lowpass as well - the input of the second scale of ScatLayerj2_f (scatternet/lowlevel.py:
 * 214-222, :255-262), which needs both s0 at full size and the first-order magnitudes from the same filtering. */
int wl_scat_fwd_level1(const void* x, void* z, void* drdx, void* drdy, void* ll, int dtype, int64_t N, int C, int H,
                       int W, const void* h0, int L0, const void* h1, int L1, int mode, double magbias,
                       int combine_colour, void* stream);

/* ScatLayerj2_f.forward (scatternet/lowlevel.py:205-295) concatenates the outputs of three transforms into its
 * (N,49,C,H/4,W/4) tensor; these two entry points let each transform write its entries in place.  Addresses in elements of
 * the dtype, for image n and channel c, q = the number of samples of an output plane: the averaged lowpass at
 * z + n z_batch_stride + z_ll_offset + c q, smoothed magnitude o (0..5) at z + n z_batch_stride + z_mag_offset + (o C + c) q.
 * z_ll_offset < 0: the averaged lowpass is not written.
 *   wl_scat_fwd_level1_into: wl_scat_fwd_level1 without combine_colour / saved tensors, q = (He/2)(We/2)
 *     (lowlevel.py:214-222 first scale, :255-262 second-order layer);
 *   wl_scat_fwd_level2_into: the second scale (:237-253: fwd_j2plus of the full-resolution lowpass x (N,C,H,W), H and W
 *     multiples of 4, then the 2x2 average of its lowpass and the magnitudes of its band-pass coefficients), q = (H/4)(W/4).
 *     Streaming kernel only (10-tap q-shift filters, planes that fill the chip): WL_ERR_UNSUPPORTED otherwise and the
 *     caller composes wl_dtcwt_fwd_level2 with its own epilogue. */
int wl_scat_fwd_level1_into(const void* x, void* z, int64_t z_batch_stride, int64_t z_ll_offset, int64_t z_mag_offset,
                            void* ll, int dtype, int64_t N, int C, int H, int W, const void* h0, int L0, const void* h1,
                            int L1, int mode, double magbias, void* stream);
int wl_scat_fwd_level2_into(const void* x, void* z, int64_t z_batch_stride, int64_t z_ll_offset, int64_t z_mag_offset,
                            int dtype, int64_t N, int C, int H, int W, const void* h0a, const void* h0b, const void* h1a,
                            const void* h1b, int L, double magbias, void* stream);

/* ScatLayer backward = ScatLayerj1_f.backward (scatternet/lowlevel.py:114-137) in ONE launch: the prologue
 * (1/4 nearest-upsample of the lowpass gradient, dr*re/r, dr*im/r) is fused into the staging of the level-1 inverse.
 * dz: gradient of z, same layout as z; drdx/drdy: as written by wl_scat_fwd_level1; dx (N,C,H,W) with H, W the
 * padded-to-even input size (callers fold the replicated edge for odd inputs, scatternet/layers.py:55-59).
 * h0/h1: the FORWARD level-1 taps.  Returns WL_ERR_UNSUPPORTED for tap counts / dtypes without a specialised
 * kernel; callers then compose the prologue themselves and call wl_dtcwt_inv_level1. */
int wl_scat_bwd_level1(const void* dz, const void* drdx, const void* drdy, void* dx, int dtype, int64_t N, int C,
                       int H, int W, const void* h0, int L0, const void* h1, int L1, int mode, int combine_colour,
                       void* stream);

/* ONE analysis level by the streaming strip kernel (csrc/wl_dwt_strip.h): the same operator as wl_dwt2d_analysis_strided
 * (AFB2D.forward, dwt/lowlevel.py:336-347) for one square filter length L (even, <= 20), float32 / float16, every mode, rows
 * of any width that are a whole number of 16-byte pieces: every input sample is read once per column strip and row
 * segment; a level whose whole row is one or two compute waves' worth of columns (128 / 256 output columns) runs four / two PLANES
 * per workgroup when the chip still gets a workgroup for every slot.  policy bit 0: 0 = the engine decides whether the launch pays
 * (workgroups for every CU; rows of 2 KiB and more, float16 256 columns, or narrower levels from 128 columns on when the planes pack
 * so that the workgroups' compute waves all have work), 1 = force;
 * bit 1 (value 2) = a HINT that each highpass bank is the quadrature mirror of its lowpass bank, h_hi[t] == (-1)^t h_lo[L-1-t]
 * (the pair of every orthogonal wavelet as stored): from 12 taps on the engine then launches a kernel variant that relies on
 * the relation - it holds the lowpass banks only - and VERIFIES it on the device against the taps as they are when it runs;
 * the two-bank variant is queued behind it and does the work when the relation does not hold (a wrong hint costs an empty
 * launch, never a wrong coefficient).
 * Returns WL_ERR_UNSUPPORTED outside its envelope (callers then use wl_dwt2d_analysis_strided). */
#define WL_TAP_SCRATCH_BYTES 64
int wl_dwt2d_analysis_stream(const void* x, int64_t x_plane_stride, int x_row_stride, void* ll, int64_t ll_plane_stride,
                             int ll_row_stride, void* highs, int dtype, int64_t planes, int H, int W, const void* h_w_lo,
                             const void* h_w_hi, const void* h_h_lo, const void* h_h_hi, int L, int mode, int policy,
                             void* stream);
/* The same with device scratch: tap_scratch = NULL, or WL_TAP_SCRATCH_BYTES of device memory that stay valid until the
 * launches of this call have run (stream order): with the hint (policy | 2) the hinted variant of 12, 14, 16 and 20 taps is
 * then the LATTICE kernel (csrc/wl_lattice.h: the column pass as K = L/2 plane rotations, half the multiplications), whose
 * coefficients a one-thread kernel derives from the column bank on the device and accepts only if they reproduce the bank to
 * 2^-22 (float32 data) / 2^-12 (float16 data) of its largest tap - else the two-bank variant runs as above.  tap_state: NULL or
 * the caller-owned record of what the block holds, exactly as for wl_dwt2d_analysis_fused_ex (the levels of one transform
 * share one examination; a launch that makes no use of the scratch - other tap counts, no hint - leaves it untouched). */
int wl_dwt2d_analysis_stream_ex(const void* x, int64_t x_plane_stride, int x_row_stride, void* ll, int64_t ll_plane_stride,
                                int ll_row_stride, void* highs, int dtype, int64_t planes, int H, int W, const void* h_w_lo,
                                const void* h_w_hi, const void* h_h_lo, const void* h_h_hi, int L, int mode, int policy,
                                void* tap_scratch, int* tap_state, void* stream);

/* ONE synthesis level by the streaming strip kernel (csrc/wl_idwt_strip.h): the same operator as wl_dwt2d_synthesis
 * (SFB2D.forward, dwt/lowlevel.py:671-680; the (OH, OW) crop is AFB2D.backward's, :356-364) for one square filter length L
 * (even, <= 20), float32 / float16, every mode, coefficient rows that are whole 16-byte pieces; highs must be present.
 * policy bit 0 as for wl_dwt2d_analysis_stream (1 = force); bit 1 (value 2) = a HINT that each highpass bank is
 * the quadrature mirror of its lowpass bank, g_hi[t] = (-1)^t g_lo[L-1-t] (the reconstruction pair of every orthogonal
 * wavelet): from 12 taps on the hinted kernel variant derives the highpass tap pairs from the lowpass ones by operand modifiers
 * instead of holding both banks in scalar registers (same arithmetic, same results) after verifying the relation on the device;
 * the two-bank variant stands by behind it as for the analysis.  Returns WL_ERR_UNSUPPORTED outside its envelope. */
int wl_dwt2d_synthesis_stream(const void* ll, int64_t ll_plane_stride, int ll_row_stride, const void* highs, void* y,
                              int dtype, int64_t planes, int Kh, int Kw, int OH, int OW, const void* g_w_lo,
                              const void* g_w_hi, const void* g_h_lo, const void* g_h_hi, int L, int mode, int policy,
                              void* stream);
/* The same with tap_scratch / tap_state as for wl_dwt2d_analysis_stream_ex (the hinted variant is then the lattice kernel, from
 * 12 taps on, 14 included). */
int wl_dwt2d_synthesis_stream_ex(const void* ll, int64_t ll_plane_stride, int ll_row_stride, const void* highs, void* y,
                                 int dtype, int64_t planes, int Kh, int Kw, int OH, int OW, const void* g_w_lo,
                                 const void* g_w_hi, const void* g_h_lo, const void* g_h_hi, int L, int mode, int policy,
                                 void* tap_scratch, int* tap_state, void* stream);

/* Gradients of the two non-separable banks, as autograd gives them upstream (where afb2d_nonsep / sfb2d_nonsep are
 * plain differentiable ATen chains, dwt/lowlevel.py:524-597, :746-798):
 *   wl_dwt2d_analysis_nonsep_bwd : dy (planes,4,Kh,Kw) -> dx (planes,H,W), the adjoint of the boundary gather + strided
 *       correlation in every mode (mirrored / wrapped / repeated samples fold their gradient back onto their source);
 *   wl_dwt2d_synthesis_nonsep_bwd: dy (planes,OH,OW) [OH = 2Kh-Ly+2, periodization 2Kh] -> dc (planes,4,Kh,Kw). */
int wl_dwt2d_analysis_nonsep_bwd(const void* dy, void* dx, int dtype, int64_t planes, int H, int W, const void* f,
                                 int Ly, int Lx, int mode, void* stream);
int wl_dwt2d_synthesis_nonsep_bwd(const void* dy, void* dc, int dtype, int64_t planes, int Kh, int Kw, const void* g,
                                  int Ly, int Lx, int mode, void* stream);

/* J (1..4) levels of the 1-D analysis bank in ONE launch (csrc/wl_dwt1d_fused.h) = DWT1DForward.forward's level loop
 * (dwt/transform1d.py:44-59 -> AFB1D.forward, dwt/lowlevel.py:368-424): x (rows,N) dense -> lo (rows,n_J) and highs[j]
 * (rows,n_{j+1}), n_{j+1} = wl_dwt_coeff_len(n_j, L, mode); every input sample is read once, the intermediate lowpass signals
 * stay in LDS.  Even L <= 20, float32 / float16, zero / symmetric / reflect for rows of any length (a workgroup owns a chunk of
 * ~4096 samples), periodic / periodization while a row fits one chunk.  Returns WL_ERR_UNSUPPORTED outside its envelope:
 * callers then chain wl_corr1d level by level. */
int wl_dwt1d_analysis_fused(const void* x, void* lo, void* const* highs, int dtype, int64_t rows, int N, int J,
                            const void* h0, const void* h1, int L, int mode, void* stream);

/* The inverse: J (1..4) levels of the 1-D synthesis bank in ONE launch (csrc/wl_idwt1d_fused.h) = DWT1DInverse.forward's level
 * loop (dwt/transform1d.py:97-115 -> SFB1D.forward, dwt/lowlevel.py:697-727, incl. the 'unpad' of a lowpass one sample longer
 * than the next highpass) and the backward of the 1-D analysis (AFB1D.backward :409-424: analysis taps, crop to the input length
 * = out_len): lo (rows,n_lo), highs[j] (rows,n_hi[j]) finest first (NULL = zeros) -> y (rows,out_len), out_len <= 2 n_hi[0] - L + 2.
 * Even L <= 20, float32 / float16, every mode but periodization.  WL_ERR_UNSUPPORTED otherwise: callers chain wl_synth1d. */
int wl_dwt1d_synthesis_fused(const void* lo, int n_lo, const void* const* highs, const int* n_hi, void* y, int out_len, int dtype,
                             int64_t rows, int J, const void* g0, const void* g1, int L, int mode, void* stream);

/* ---- single-axis building blocks -------------------------------------------------------------------------------
 * One strided / dilated correlation with boundary extension along the middle axis of a dense (outer, n, inner) tensor:
 *   y[o, out_offset + out_stride*k, i] = sum_{t<ntaps} h[tap_offset + tap_stride*t] * ext(x[o,:,i], start + step*k + tap_step*t)
 * for k in [0,K); y index = o*y_outer_stride + q*inner + i.  h1/y1 (nullable) = a second tap set / output on the same
 * samples.  `ext`: 0 zero, 1 symmetric (half-sample), 2 reflect (whole-sample), 3 periodic, 4 periodization, 5 replicate,
 * 6 periodization exactly as the reference evaluates it (roll, zero-padded strided convolution, ONE fold of the wrapped tail,
 * dwt/lowlevel.py:134-150; differs from 4 when the signal is shorter than the filter; step 2, tap_step 1, start 1-ntaps).
 * Replaces the ATen bodies of afb1d on one axis (AFB1D.forward, dwt/lowlevel.py:368-407 -> :91-172), afb1d_atrous
 * (:175-223) and the DTCWT primitives colfilter / rowfilter / coldfilt / rowdfilt / colifilt / rowifilt
 * (dtcwt/lowlevel.py:70-239), each of which is one to four such correlations. */
int wl_corr1d(const void* x, void* y0, void* y1, int dtype, int64_t outer, int n, int64_t inner, int64_t y_outer_stride,
              int K, const void* h0, const void* h1, int tap_offset, int tap_stride, int ntaps, int start, int step,
              int tap_step, int ext, int out_offset, int out_stride, void* stream);

/* nlev (1..4) analysis levels of SMALL planes (up to about 72 x 72: CNN feature maps, CIFAR-sized images) in ONE launch with
 * several planes per workgroup (csrc/wl_dwt_small.h) = the level loop of DWTForward.forward (dwt/transform2d.py:63-74) where
 * the streaming kernel of wl_dwt2d_analysis_fused would give each 32 x 32 plane a workgroup of its own.  Same contract as
 * wl_dwt2d_analysis_fused (x (planes,H,W) dense -> yl, yh[j] (planes,3,Kh_j,Kw_j), sizes by wl_dwt_coeff_len); every mode,
 * up to 20 taps, any sizes, float32 / float16.  WL_ERR_UNSUPPORTED: larger planes, float64, a periodization level shorter
 * than the filter. */
int wl_dwt2d_analysis_small(const void* x, void* yl, void* const* yh, int dtype, int64_t planes, int H, int W, int nlev,
                            const void* h_w_lo, const void* h_w_hi, const void* h_h_lo, const void* h_h_hi, int L, int mode,
                            void* stream);

/* The inverse of the above: nlev (1..4) synthesis levels of small planes in ONE launch, several planes per workgroup
 * (csrc/wl_dwt_small.h) = the level loop of DWTInverse.forward (dwt/transform2d.py:131-148 incl. the 'unpad' crop of
 * :141-146) and the backward of the analysis (AFB2D.backward, dwt/lowlevel.py:350-365, with the analysis taps).  yl
 * (planes,yl_h,yl_w) dense, yh[j] (planes,3,Kh[j],Kw[j]) finest first (NULL = zeros), y (planes,OH,OW) with OH = 2 Kh[0] - L + 2
 * (2 Kh[0] for periodization); the caller crops.  Same envelope and declines as wl_dwt2d_analysis_small; even L. */
int wl_dwt2d_synthesis_small(const void* yl, int yl_h, int yl_w, const void* const* yh, const int* Kh, const int* Kw, void* y,
                             int dtype, int64_t planes, int nlev, const void* g_w_lo, const void* g_w_hi, const void* g_h_lo,
                             const void* g_h_hi, int L, int mode, void* stream);

/* Level 1 of the rotationally symmetric DTCWT (biort 'near_sym_b_bp': a third, band-pass pair for the diagonal sub-band)
 * in ONE launch (csrc/wl_dtcwt_rot.h) = fwd_j1_rot (dtcwt/transform_funcs.py:124-149: three rowfilter, four colfilter,
 * three q2c, two stack), orientations along dim 1 as ScatLayerj1_rot_f uses it (scatternet/lowlevel.py:140-182):
 *   scat == 0: x (N,C,H,W) -> ll (N,C,H,W), re / im (N,6,C,H/2,W/2);
 *   scat == 1: x -> re = Z (N,7,C,H/2,W/2): Z[:,0] = the 2x2 average of ll, Z[:,1+o] = sqrt(re_o^2 + im_o^2 + b^2) - b
 *              (ScatLayerj1_rot_f.forward without combine_colour; ll and im unused).
 * h0 / h1 / h2: odd lengths up to 19; mode 1 = symmetric extension, anything else zero padding (dtcwt/lowlevel.py:75-79);
 * even H and W (the modules pad first), else WL_ERR_UNSUPPORTED: callers then chain wl_corr1d. */
int wl_dtcwt_fwd_level1_rot(const void* x, void* ll, void* re, void* im, int dtype, int64_t N, int C, int H, int W,
                            const void* h0, int L0, const void* h1, int L1, const void* h2, int L2, int mode,
                            int scat, double magbias, void* stream);

/* One level of the 2-D stationary (undecimated) transform in ONE launch (csrc/wl_swt2d.h) = afb2d_atrous
 * (dwt/lowlevel.py:475-521: afb1d_atrous :175-223 along W, then along H; the level of SWTForward.forward,
 * dwt/transform2d.py:186-212): x (planes,H,W) through the plane stride x_ps (elements; rows dense - the ll channels of a
 * previous level are every 4th plane of its output) -> y (planes,4,H,W) dense with sub-band 2 r + b (r: band along W,
 * b: band along H) = the (N,4C,H,W) tensor the reference returns.  Taps: the stored (reversed) filters, Lw / Lh of them,
 * dilated by `dilation`; `ext` as for wl_corr1d (0 zero, 1 symmetric, 2 reflect, 3 periodic, 5 replicate).  Returns
 * WL_ERR_UNSUPPORTED for odd L * dilation and for dilated filters too long for a tile in LDS: callers then chain wl_corr1d. */
int wl_swt2d_level(const void* x, int64_t x_ps, void* y, int dtype, int64_t planes, int H, int W,
                   const void* h_w_lo, const void* h_w_hi, const void* h_h_lo, const void* h_h_hi, int Lw, int Lh,
                   int dilation, int ext, void* stream);

/* 1-D two-channel synthesis bank along the middle axis = sfb1d (dwt/lowlevel.py:226-271; SFB1D.forward :697-727 and
 * AFB1D.backward :409-424): lo, hi (outer, K, inner) [hi may be NULL = zeros] -> y (outer, ny, inner) with
 * ny <= 2K-L+2 (2K for periodization, whose single fold of the wrapped tail is reproduced literally). */
int wl_synth1d(const void* lo, const void* hi, void* y, int dtype, int64_t outer, int K, int64_t inner, int ny,
               const void* g0, const void* g1, int L, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAVELETS_HIP_H */
