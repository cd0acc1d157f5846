"""ctypes prototypes for the C ABI declared in include/wavelets_hip.h."""
import ctypes as C

P = C.c_void_p
I = C.c_int
L = C.c_int64

PROTOTYPES = {
    'wl_version': (I, []),
    'wl_backend': (C.c_char_p, []),
    'wl_set_option': (I, [C.c_char_p, I]),
    'wl_get_option': (I, [C.c_char_p]),
    'wl_last_kernel': (C.c_char_p, []),
    'wl_launch_count': (C.c_longlong, []),
    'wl_last_grid': (C.c_longlong, []),
    'wl_kernel_history': (C.c_char_p, [I]),
    'wl_dwt_coeff_len': (I, [I, I, I]),
    'wl_dwt2d_analysis': (I, [P, P, P, I, L, I, I, P, P, I, P, P, I, I, P]),
    'wl_dwt2d_analysis_strided': (I, [P, L, I, P, L, I, P, I, L, I, I, P, P, I, P, P, I, I, P]),
    'wl_dwt2d_synthesis': (I, [P, L, I, P, P, I, L, I, I, I, I, P, P, I, P, P, I, I, P]),
    'wl_dwt2d_analysis_fused': (I, [P, P, C.POINTER(P), I, L, I, I, I, P, P, P, P, I, I, I, P]),
    'wl_dwt2d_analysis_fused_ex': (I, [P, L, I, P, C.POINTER(P), I, L, I, I, I, P, P, P, P, I, I, I, P, C.POINTER(I), P]),
    'wl_dwt2d_synthesis_fused': (I, [P, L, I, I, I, C.POINTER(P), C.POINTER(I), C.POINTER(I), P, I, L, I, P, P, P, P, I, I, I, P]),
    'wl_dwt2d_synthesis_fused_ex': (I, [P, L, I, I, I, C.POINTER(P), C.POINTER(I), C.POINTER(I), P, I, L, I, P, P, P, P, I, I, I, P, C.POINTER(I), P]),
    'wl_dwt2d_analysis_stream': (I, [P, L, I, P, L, I, P, I, L, I, I, P, P, P, P, I, I, I, P]),
    'wl_dwt2d_analysis_stream_ex': (I, [P, L, I, P, L, I, P, I, L, I, I, P, P, P, P, I, I, I, P, C.POINTER(I), P]),
    'wl_dwt2d_synthesis_stream': (I, [P, L, I, P, P, I, L, I, I, I, I, P, P, P, P, I, I, I, P]),
    'wl_dwt2d_synthesis_stream_ex': (I, [P, L, I, P, P, I, L, I, I, I, I, P, P, P, P, I, I, I, P, C.POINTER(I), P]),
    'wl_dwt2d_analysis_nonsep': (I, [P, P, I, L, I, I, P, I, I, I, P]),
    'wl_dwt2d_synthesis_nonsep': (I, [P, P, I, L, I, I, I, I, P, I, I, I, P]),
    'wl_dwt2d_analysis_nonsep_bwd': (I, [P, P, I, L, I, I, P, I, I, I, P]),
    'wl_dwt2d_synthesis_nonsep_bwd': (I, [P, P, I, L, I, I, P, I, I, I, P]),
    'wl_dtcwt_fwd_level1': (I, [P, P, P, I, L, I, I, P, I, P, I, I, P]),
    'wl_dtcwt_fwd_level2': (I, [P, P, P, I, L, I, I, P, P, P, P, I, P]),
    'wl_dtcwt_fwd_level12': (I, [P, P, P, P, I, L, I, I, P, I, P, I, P, P, P, P, I, I, I, P]),
    'wl_dtcwt_inv_level21': (I, [P, L, I, P, P, P, I, L, I, I, P, I, P, I, P, P, P, P, I, I, I, P]),
    'wl_dtcwt_inv_level1': (I, [P, L, I, P, P, I, L, I, I, P, I, P, I, I, P]),
    'wl_dtcwt_inv_level2': (I, [P, L, I, P, P, I, L, I, I, P, P, P, P, I, P]),
    'wl_scat_fwd_level1': (I, [P, P, P, P, P, I, L, I, I, I, P, I, P, I, I, C.c_double, I, P]),
    'wl_scat_fwd_level1_into': (I, [P, P, L, L, L, P, I, L, I, I, I, P, I, P, I, I, C.c_double, P]),
    'wl_scat_fwd_level2_into': (I, [P, P, L, L, L, I, L, I, I, I, P, P, P, P, I, C.c_double, P]),
    'wl_dwt1d_analysis_fused': (I, [P, P, C.POINTER(P), I, L, I, I, P, P, I, I, P]),
    'wl_dwt2d_analysis_small': (I, [P, P, C.POINTER(P), I, L, I, I, I, P, P, P, P, I, I, P]),
    'wl_dwt2d_synthesis_small': (I, [P, I, I, C.POINTER(P), C.POINTER(I), C.POINTER(I), P, I, L, I, P, P, P, P, I, I, P]),
    'wl_dtcwt_fwd_level1_rot': (I, [P, P, P, P, I, L, I, I, I, P, I, P, I, P, I, I, I, C.c_double, P]),
    'wl_swt2d_level': (I, [P, L, P, I, L, I, I, P, P, P, P, I, I, I, I, P]),
    'wl_dwt1d_synthesis_fused': (I, [P, I, C.POINTER(P), C.POINTER(I), P, I, I, L, I, P, P, I, I, P]),
    'wl_corr1d': (I, [P, P, P, I, L, I, L, L, I, P, P, I, I, I, I, I, I, I, I, I, P]),
    'wl_synth1d': (I, [P, P, P, I, L, I, L, I, P, P, I, I, P]),
    'wl_scat_bwd_level1': (I, [P, P, P, P, I, L, I, I, I, P, I, P, I, I, I, P]),
}


def bind(lib):
    """Attach argtypes/restype for every exported entry point; raises AttributeError when the
    library does not export a symbol the header declares."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
