"""ctypes binding of the C-ABI in include/asr_amd.h.  Loading fails loudly: there is no fallback
path -- every op of the package runs through libasr_amd.so."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ASR_AMD_LIB') or os.path.join(_HERE, 'libasr_amd.so')

ERR = {0: 'ASR_OK', -1: 'ASR_ERR_ARG', -2: 'ASR_ERR_LAUNCH', -3: 'ASR_ERR_WS', -4: 'ASR_ERR_TIMEOUT',
       -5: 'ASR_ERR_LIMIT'}

fp = C.c_void_p          # device pointers travel as integers
i32 = C.c_int
f32 = C.c_float
sz = C.c_size_t


class LstmDir(C.Structure):
    _fields_ = [('reverse', i32), ('xin', fp), ('wpack', fp), ('alpha_hid', fp), ('beta_hid', fp),
                ('alpha_c', fp), ('beta_c', fp), ('c0', fp), ('r0', fp), ('a0', fp), ('r_out', fp), ('ld_r', i32),
                ('gates', fp), ('cst', fp), ('ahat', fp), ('rs_a', fp), ('chat', fp), ('rs_c', fp),
                ('dr_ext', fp), ('ld_dr', i32), ('dG', fp), ('dA', fp), ('dOut', fp),
                ('peep', fp), ('a0_until_valid', i32), ('dc_ext', fp), ('ld_dc', i32), ('dc0', fp), ('dr0', fp),
                ('da0', fp), ('dpeep', fp)]


class LstmScanDesc(C.Structure):
    _fields_ = [('T', i32), ('B', i32), ('N', i32), ('ndir', i32), ('ln_kind', i32), ('cell_ln', i32),
                ('grad_clip', f32), ('mask', fp), ('mask_stride_b', i32), ('mask_stride_t', i32),
                ('dir', LstmDir * 2)]


class HyperDir(C.Structure):
    _fields_ = [('reverse', i32), ('xproj', fp), ('xhproj', fp), ('wpack', fp),
                ('cvec', fp), ('r_out', fp), ('ld_r', i32), ('hh_out', fp),
                ('gates', fp), ('cst', fp), ('hgates', fp), ('hcst', fp), ('aD', fp),
                ('dr_ext', fp), ('ld_dr', i32), ('dXp', fp), ('dXh', fp), ('dA', fp), ('dD', fp),
                ('dcv', fp), ('dbh', fp), ('carry', fp)]


class PrjHyperDir(C.Structure):
    _fields_ = [('reverse', i32)] + [(n, fp) for n in (
        'inner_in', 'outer_in', 'Wf', 'W3', 's', 'ai', 'bi_c', 'ao', 'bo_c', 'r',
        'gi', 'go', 'ci', 'co', 'xhi', 'xho', 'rsi', 'rso', 'sc', 'a',
        'dr', 'dta', 'doi', 'dsc', 'douti', 'douto')]


class PrjHyperDesc(C.Structure):
    _fields_ = [('T', i32), ('B', i32), ('N', i32), ('F', i32), ('ndir', i32), ('b0', i32), ('Btot', i32),
                ('grad_clip', f32), ('mask', fp), ('mask_stride_b', i32), ('mask_stride_t', i32),
                ('dir', PrjHyperDir * 2)]


class ScalingDir(C.Structure):
    _fields_ = [('reverse', i32)] + [(n, fp) for n in (
        'inner_in', 'outer_in', 'Wio', 'S', 's', 'Woo', 'wpeep', 'lna_i', 'lnb_i', 'lna_o', 'lnb_o', 'init', 'hid',
        'gi', 'go', 'ci', 'co', 'xh_i', 'xh_o', 'rs_i', 'rs_o', 'sc', 'a',
        'dhid', 'dgi', 'dgo', 'dsc', 'da', 'dh', 'dy_i', 'dy_o', 'dpeep', 'dinit', 'tg0', 'dgi0')]


class ScalingDesc(C.Structure):
    _fields_ = [('T', i32), ('B', i32), ('N', i32), ('H', i32), ('ndir', i32), ('b0', i32), ('Btot', i32),
                ('use_peepholes', i32), ('use_layer_norm', i32), ('grad_clip', f32), ('mask', fp),
                ('mask_stride_b', i32), ('mask_stride_t', i32), ('dir', ScalingDir * 2)]


class HyperScanDesc(C.Structure):
    _fields_ = [('T', i32), ('B', i32), ('N', i32), ('H', i32), ('ndir', i32), ('step0', i32), ('step1', i32),
                ('grad_clip', f32), ('clip_outcell', i32), ('lhuc_reparam', i32), ('mask', fp), ('mask_stride_b', i32), ('mask_stride_t', i32),
                ('dir', HyperDir * 2)]


_SIGS = {
    'asr_gemm_f32': (i32, [i32, i32, i32, i32, i32, f32, fp, i32, fp, i32, f32, fp, i32, fp, fp]),
    'asr_gemm_f32_ws': (i32, [i32, i32, i32, i32, i32, f32, fp, i32, fp, i32, f32, fp, i32, fp, fp, sz, fp]),
    'asr_gemm_bf16_ws': (i32, [i32, i32, i32, i32, i32, f32, fp, i32, fp, i32, f32, fp, i32, fp, fp, sz, fp]),
    'asr_gemm_bf16x3_ws': (i32, [i32, i32, i32, i32, i32, i32, f32, fp, i32, fp, i32, f32, fp, i32, fp, fp, sz, fp]),
    'asr_split_ld': (i32, [i32]),
    'asr_split_bf16x3': (i32, [fp, i32, i32, i32, fp, i32, sz, fp]),
    'asr_gemm_presplit_ws': (i32, [i32, i32, i32, i32, i32, i32, i32, f32, fp, i32, sz, fp, i32, sz, f32, fp, i32, fp, fp, sz, fp]),
    'asr_swap01': (i32, [fp, fp, i32, i32, i32, i32, i32, fp]),
    'asr_ln_rows_fwd': (i32, [fp, i32, i32, fp, i32, fp, fp, i32, fp, fp, fp, fp, i32, fp]),
    'asr_ln_rows_bwd': (i32, [fp, i32, i32, fp, fp, fp, i32, fp, i32, fp]),
    'asr_colreduce_ws_bytes': (sz, [i32, i32]),
    'asr_colreduce': (i32, [fp, i32, fp, i32, i32, i32, i32, fp, i32, fp, sz, fp]),
    'asr_colreduce_static': (i32, [fp, i32, fp, i32, i32, i32, i32, fp, i32, fp]),
    'asr_bias_act': (i32, [fp, fp, i32, i32, i32, fp]),
    'asr_act_bwd': (i32, [fp, fp, i32, i32, i32, fp]),
    'asr_act_eval': (i32, [i32, fp, fp, sz, fp]),
    'asr_rowscale': (i32, [fp, i32, fp, i32, i32, fp, i32, i32, i32, fp]),
    'asr_dropout': (i32, [fp, i32, fp, i32, i32, i32, f32, C.c_ulonglong, C.c_ulonglong, fp]),
    'asr_affine_rows': (i32, [fp, i32, fp, fp, fp, fp, i32, i32, i32, fp, i32, i32, i32, fp]),
    'asr_var_costs': (i32, [fp, fp, fp, i32, i32, i32, f32, f32, i32, i32, fp, fp, i32, fp]),
    'asr_masked_mean': (i32, [fp, fp, i32, i32, i32, fp, fp]),
    'asr_masked_mean_tanh_bwd': (i32, [fp, fp, fp, i32, i32, i32, fp, fp]),
    'asr_seq_pool_fwd': (i32, [fp, sz, sz, fp, i32, i32, i32, i32, fp, fp]),
    'asr_seq_pool_bwd': (i32, [fp, fp, fp, sz, sz, fp, i32, i32, i32, i32, fp, fp]),
    'asr_softmax_ce': (i32, [fp, i32, i32, i32, fp, i32, i32, fp, i32, i32, f32, fp, fp, fp, fp, i32,
                             fp, fp]),
    'asr_reduce_sum': (i32, [fp, sz, fp, i32, fp]),
    'asr_adam_step': (i32, [fp, fp, fp, fp, sz, f32, f32, f32, f32, fp, i32, fp, fp]),
    'asr_sqsum': (i32, [fp, sz, fp, i32, fp, fp]),
    'asr_prjhyper_ws_bytes': (sz, [i32, i32, i32, i32]),
    'asr_prjhyper_fwd': (i32, [C.POINTER(PrjHyperDesc), fp, sz, fp]),
    'asr_prjhyper_bwd': (i32, [C.POINTER(PrjHyperDesc), fp, sz, fp]),
    'asr_prjhyper_set_mode': (i32, [i32]),
    'asr_scaling_ws_bytes': (sz, [i32, i32, i32, i32]),
    'asr_scaling_fwd': (i32, [C.POINTER(ScalingDesc), fp, sz, fp]),
    'asr_scaling_bwd': (i32, [C.POINTER(ScalingDesc), fp, sz, fp]),
    'asr_scaling_set_mode': (i32, [i32]),
    'asr_opt_step': (i32, [i32, fp, fp, fp, fp, sz, f32, f32, f32, f32, fp, fp]),
    'asr_norm_clip_mult': (i32, [fp, f32, f32, fp, fp]),
    'asr_lstm_pack_floats': (sz, [i32, i32]),
    'asr_lstm_pack_w': (i32, [fp, i32, i32, fp, fp]),
    'asr_lstm_bwd_pack_mode': (i32, [i32, i32, i32]),
    'asr_lstm_scan_ws_bytes': (sz, [i32, i32, i32, i32, i32]),
    'asr_lstm_scan_fwd': (i32, [C.POINTER(LstmScanDesc), fp, sz, fp]),
    'asr_lstm_scan_bwd': (i32, [C.POINTER(LstmScanDesc), fp, sz, fp]),
    'asr_scan_status': (i32, [fp, i32]),
    'asr_scan_set_chunks_per_wg': (i32, [i32]),
    'asr_perm4': (i32, [fp, fp, i32, i32, i32, fp]),
    'asr_hyper_pack_floats': (sz, [i32, i32, i32]),
    'asr_hyper_pack_w': (i32, [fp, fp, fp, fp, i32, i32, i32, fp, fp]),
    'asr_hyper_scan_ws_bytes': (sz, [i32, i32, i32, i32, i32, i32]),
    'asr_hyper_scan_fwd': (i32, [C.POINTER(HyperScanDesc), fp, sz, fp]),
    'asr_hyper_scan_bwd': (i32, [C.POINTER(HyperScanDesc), fp, sz, fp]),
}

_lib = None


def declared_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('asr_amd: %s is missing -- build it with `python -m asr_amd.build` '
                               '(there is no CPU fallback)' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)      # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class AsrError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise AsrError('%s failed: %s (%d)' % (what, ERR.get(rc, '?'), rc))
