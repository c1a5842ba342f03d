"""ctypes binding of libestts.so (the C ABI declared in include/estts.h).

There is no fallback: if the shared library is missing or does not export a
symbol, importing an op raises.  Build it with `make -C <package>/csrc` (or
`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libestts.so')

ABI_VERSION = 36
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
ACT = {None: ACT_NONE, 'linear': ACT_NONE, 'none': ACT_NONE, 'relu': ACT_RELU, 'tanh': ACT_TANH}

_vp, _i32, _f32, _u32, _i64 = C.c_void_p, C.c_int32, C.c_float, C.c_uint32, C.c_int64


class GemmArgs(C.Structure):
    _fields_ = [('A', _vp), ('lda', _i32), ('seg_len', _i32), ('taps', _i32), ('c_per_tap', _i32), ('tap_lpad', _i32),
                ('W_hi', _vp), ('W_lo', _vp), ('ldw', _i32), ('M', _i32), ('N', _i32), ('K', _i32),
                ('s1', _vp), ('b1', _vp), ('act', _i32), ('alpha', _f32),
                ('R', _vp), ('ldr', _i32), ('r_mod', _i32), ('s2', _vp), ('b2', _vp),
                ('C', _vp), ('ldc', _i32), ('precision', _i32),
                ('ws', _vp), ('ws_bytes', C.c_int64), ('counters', _vp), ('n_counters', _i32), ('tune', _i32),
                ('conv_h', _i32), ('conv_w', _i32), ('conv_wo', _i32), ('conv_howo', _i32), ('conv_kw', _i32),
                ('conv_stride', _i32), ('conv_pt', _i32), ('conv_pl', _i32),
                ('drop_rate', _f32), ('drop_seed', _u32), ('drop_site', _u32), ('drop_row0', _i32),
                ('batch', _i32), ('batch_stride_a', _i64), ('batch_stride_w', _i64), ('batch_stride_c', _i64),
                ('batch_stride_v', _i64),
                ('ln_y', _vp), ('ln_x', _vp), ('ln_ga', _vp), ('ln_ba', _vp), ('ln_gb', _vp), ('ln_bb', _vp), ('ln_out', _vp),
                ('ln_ldy', _i32), ('ln_ldx', _i32), ('ln_ldo', _i32), ('ln_act', _i32), ('ln_eps', _f32),
                ('stat_slots', _vp), ('stat_shift', _vp), ('ln_epilogue', _i32)]


class AttentionArgs(C.Structure):
    _fields_ = [('q', _vp), ('ldq', _i32), ('k', _vp), ('ldk', _i32), ('v', _vp), ('ldv', _i32),
                ('ctx', _vp), ('ldo', _i32), ('attn', _vp), ('key_mask', _vp),
                ('B', _i32), ('H', _i32), ('Lq', _i32), ('Lk', _i32), ('depth', _i32),
                ('causal', _i32), ('precision', _i32), ('kv_shared', _i32), ('variant', _i32)]


class RownormArgs(C.Structure):
    _fields_ = [('y', _vp), ('ldy', _i32), ('x', _vp), ('ldx', _i32), ('ga', _vp), ('ba', _vp), ('gb', _vp), ('bb', _vp),
                ('act', _i32), ('eps', _f32), ('out', _vp), ('ldo', _i32), ('M', _i32), ('D', _i32),
                ('dot_w', _vp), ('dot_b', _f32), ('dot_scale', _f32), ('dot_mask', _vp), ('dot_out', _vp),
                ('drop_rate', _f32), ('drop_seed', _u32), ('drop_site', _u32), ('dot_b_ptr', _vp)]


class RownormBwdArgs(C.Structure):
    _fields_ = [('f', RownormArgs), ('dout', _vp), ('lddout', _i32), ('ddot', _vp), ('dy', _vp), ('lddy', _i32),
                ('dx', _vp), ('lddx', _i32), ('dx_accumulate', _i32), ('dga', _vp), ('dba', _vp), ('dgb', _vp), ('dbb', _vp),
                ('ddot_w', _vp), ('ddot_b', _vp), ('ws', _vp), ('ws_bytes', _i64)]


class WgradArgs(C.Structure):
    _fields_ = [('A', _vp), ('lda', _i32), ('seg_len', _i32), ('taps', _i32), ('c_per_tap', _i32), ('tap_lpad', _i32),
                ('dY', _vp), ('lddy', _i32), ('M', _i32), ('N', _i32), ('K', _i32), ('dW', _vp), ('lddw', _i32),
                ('precision', _i32), ('splits', _i32), ('db', _vp)]


class AttentionBwdArgs(C.Structure):
    _fields_ = [('q', _vp), ('ldq', _i32), ('k', _vp), ('ldk', _i32), ('v', _vp), ('ldv', _i32), ('ctx', _vp), ('ldo', _i32),
                ('dctx', _vp), ('lddo', _i32), ('attn', _vp), ('delta', _vp), ('dq', _vp), ('lddq', _i32),
                ('dk', _vp), ('lddk', _i32), ('dv', _vp), ('lddv', _i32),
                ('B', _i32), ('H', _i32), ('Lq', _i32), ('Lk', _i32), ('depth', _i32), ('precision', _i32)]


class WavernnPrepBwdArgs(C.Structure):
    _fields_ = [('dxin', _vp), ('ld_dxin', _i32), ('da2', _vp), ('ld_da2', _i32), ('da3', _vp), ('ld_da3', _i32),
                ('da4', _vp), ('ld_da4', _i32), ('mel', _vp), ('mel_off', _vp), ('d_aux', _vp), ('dfir', _vp),
                ('ntaps', _i32), ('fir_kmin', _i32), ('n_utt', _i32), ('frames', _i32), ('feat', _i32), ('d', _i32),
                ('hop', _i32), ('pad', _i32)]


class RowmaskArgs(C.Structure):
    _fields_ = [('x', _vp), ('ldx', _i32), ('mask', _vp), ('M', _i32), ('D', _i32)]


PROG_MAX_PATCH = 8


class CrossRowArgs(C.Structure):
    _fields_ = [('x', _vp), ('ldx', _i32), ('ln_y', _vp), ('ln_x', _vp), ('ln_ga', _vp), ('ln_ba', _vp), ('ln_gb', _vp), ('ln_bb', _vp),
                ('ln_out', _vp), ('ln_ldy', _i32), ('ln_ldx', _i32), ('ln_ldo', _i32), ('ln_act', _i32), ('ln_eps', _f32),
                ('wq_hi', _vp), ('wq_lo', _vp), ('ldwq', _i32), ('bq', _vp),
                ('k', _vp), ('v', _vp), ('ldk', _i32), ('ldv', _i32), ('kv_batch_stride', _i64), ('key_mask', _vp),
                ('attn', _vp), ('attn_row_stride', _i64),
                ('wo_hi', _vp), ('wo_lo', _vp), ('ldwo', _i32), ('bo', _vp),
                ('y', _vp), ('ldy', _i32), ('M', _i32), ('d', _i32), ('H', _i32), ('Lk', _i32),
                ('ws', _vp), ('ws_bytes', _i64), ('counters', _vp), ('n_counters', _i32), ('q', _vp), ('ldq', _i32)]


class DecoderBlockArgs(C.Structure):
    _fields_ = [('x', _vp), ('ldx', _i32), ('ln_y', _vp), ('ln_x', _vp), ('ln_ga', _vp), ('ln_ba', _vp), ('ln_gb', _vp), ('ln_bb', _vp),
                ('ln_ldy', _i32), ('ln_ldx', _i32), ('ln_act', _i32), ('ln_eps', _f32),
                ('wqkv_hi', _vp), ('wqkv_lo', _vp), ('ldwqkv', _i32), ('bqkv', _vp),
                ('qkv', _vp), ('ldqkv', _i32), ('qkv_batch_stride', _i64), ('pos', _i32),
                ('self_mask', _vp), ('self_mask_stride', _i64),
                ('wos_hi', _vp), ('wos_lo', _vp), ('ldwos', _i32), ('bos', _vp),
                ('sa_ga', _vp), ('sa_ba', _vp), ('sa_gb', _vp), ('sa_bb', _vp),
                ('wqc_hi', _vp), ('wqc_lo', _vp), ('ldwqc', _i32), ('bqc', _vp),
                ('k', _vp), ('v', _vp), ('ldk', _i32), ('ldv', _i32), ('kv_batch_stride', _i64), ('Lk', _i32), ('key_mask', _vp),
                ('attn', _vp), ('attn_row_stride', _i64),
                ('woc_hi', _vp), ('woc_lo', _vp), ('ldwoc', _i32), ('boc', _vp), ('ca_g', _vp), ('ca_b', _vp),
                ('wff_hi', _vp), ('wff_lo', _vp), ('ldwff', _i32), ('bff', _vp),
                ('y2', _vp), ('ldy2', _i32), ('x2', _vp), ('ldx2', _i32),
                ('ws', _vp), ('ws_bytes', _i64), ('counters', _vp), ('n_counters', _i32), ('M', _i32), ('d', _i32), ('H', _i32)]


class PrenetRowArgs(C.Structure):
    _fields_ = [('x', _vp), ('ldx', _i32), ('C', _i32), ('mask', _vp),
                ('w1_hi', _vp), ('w1_lo', _vp), ('ldw1', _i32), ('b1', _vp), ('N1', _i32),
                ('w2_hi', _vp), ('w2_lo', _vp), ('ldw2', _i32), ('b2', _vp), ('N2', _i32),
                ('alpha', _f32), ('R', _vp), ('ldr', _i32),
                ('drop_rate', _f32), ('drop_seed', _u32), ('drop_site1', _u32), ('drop_site2', _u32), ('drop_row0', _i32),
                ('out', _vp), ('ldo', _i32), ('M', _i32)]


class ProgOp(C.Structure):
    _fields_ = [('kind', _i32), ('n_patch', _i32), ('args', _vp), ('patch_offset', _i32 * PROG_MAX_PATCH),
                ('patch_is_ptr', _i32 * PROG_MAX_PATCH), ('patch_stride', _i64 * PROG_MAX_PATCH)]


class PackJob(C.Structure):
    _fields_ = [('src', _vp), ('hi', _vp), ('lo', _vp), ('first_unit', _i64), ('ld_src', _i32), ('K', _i32), ('N', _i32),
                ('taps', _i32), ('mode', _i32), ('pad_', _i32)]


class ColreduceArgs(C.Structure):
    _fields_ = [('z', _vp), ('ldz', _i32), ('g', _vp), ('ldg', _i32), ('a', _vp), ('lda', _i32), ('act', _i32),
                ('mean', _vp), ('rstd', _vp), ('out0', _vp), ('out1', _vp), ('dy_out', _vp), ('lddy', _i32),
                ('M', _i32), ('C', _i32), ('mode', _i32), ('scale', _vp), ('shift', _vp), ('ws', _vp), ('ws_bytes', _i64),
                ('defer', _i32), ('clear', _vp), ('clear_floats', _i64)]


class GruFusedArgs(C.Structure):
    _fields_ = [('gi', _vp), ('ldgi', _i32), ('gh', _vp), ('ldgh', _i32), ('h_prev', _vp), ('ldhp', _i32),
                ('W_hi', _vp), ('W_lo', _vp), ('ldw', _i32), ('bias', _vp), ('h_out', _vp), ('ldho', _i32),
                ('x_res', _vp), ('ldx', _i32), ('out', _vp), ('ldo', _i32), ('dgh_next', _vp), ('lddghn', _i32),
                ('dh', _vp), ('lddh', _i32), ('dgi', _vp), ('lddgi', _i32), ('dgh', _vp), ('lddgh', _i32),
                ('dhd', _vp), ('lddhd', _i32), ('M', _i32), ('H', _i32), ('precision', _i32)]


class WavernnPrepArgs(C.Structure):
    _fields_ = [('mel', _vp), ('mel_off', _vp), ('aux', _vp), ('aux_off', _vp), ('fir', _vp), ('ntaps', _i32),
                ('fir_kmin', _i32), ('seq_utt', _vp), ('seq_start', _vp), ('utt_samples', _vp), ('x_prev', _vp),
                ('step', _vp), ('xin', _vp), ('ld_xin', _i32), ('a2', _vp), ('ld_a2', _i32), ('a3', _vp), ('ld_a3', _i32),
                ('a4', _vp), ('ld_a4', _i32), ('n_seq', _i32), ('feat', _i32), ('d', _i32), ('hop', _i32), ('pad', _i32)]


class WavernnLayerArgs(C.Structure):
    _fields_ = [('a0', _vp), ('ks0', _i32), ('kpitch0', _i32), ('a1', _vp), ('ks1', _i32), ('kpitch1', _i32), ('W', _vp),
                ('bias', _vp), ('n_seq', _i32), ('n_units', _i32), ('mode', _i32), ('act', _i32), ('unit_blocks', _i32),
                ('seq_tile', _i32),
                ('h_prev', _vp), ('ldh', _i32), ('x_res', _vp), ('ldr', _i32), ('h_out', _vp), ('ldho', _i32),
                ('h_panel', _vp), ('h_kpitch', _i32), ('out_f32', _vp), ('ldo', _i32), ('out_panel', _vp), ('out_kpitch', _i32)]


class WavernnCondArgs(C.Structure):
    _fields_ = [('mel', _vp), ('aux', _vp), ('fir', _vp), ('ntaps', _i32), ('seq_info', _vp), ('step', _vp), ('done', _vp),
                ('xin_panel', _vp), ('a2_panel', _vp), ('a3_panel', _vp), ('a4_panel', _vp),
                ('aux_ks', _i32), ('aux_kpitch', _i32), ('n_seq', _i32), ('feat', _i32), ('d', _i32), ('hop', _i32)]


class WavernnUnfoldArgs(C.Structure):
    _fields_ = [('wave', _vp), ('ld_wave', _i32), ('seq_off', _vp), ('n_folds', _vp), ('out_len', _vp), ('out_off', _vp),
                ('out', _vp), ('n_utt', _i32), ('target', _i32), ('overlap', _i32), ('max_len', _i32), ('mu', _i32),
                ('fade_tail', _i32), ('step_in', C.c_double), ('step_tail', C.c_double)]


class WavernnSampleArgs(C.Structure):
    _fields_ = [('logits', _vp), ('ld_logits', _i32), ('n_classes', _i32), ('mode', _i32), ('u', _vp), ('u_mix', _vp),
                ('forced', _vp), ('seed', C.c_uint32), ('wave', _vp), ('ld_wave', _i32), ('x_prev', _vp), ('step', _vp),
                ('done', _vp), ('step_next', _vp), ('n_seq', _i32), ('with_cond', _i32), ('cond', WavernnCondArgs)]





# symbol -> (restype, argtypes); every symbol declared in include/estts.h must be listed here
SIGNATURES = {
    'estts_abi_version': (_i32, []),
    'estts_last_error': (C.c_char_p, []),
    'estts_launch_count': (C.c_longlong, []),
    'estts_gemm': (_i32, [C.POINTER(GemmArgs), _vp]),
    'estts_gemm_noalloc': (_i32, [C.POINTER(GemmArgs), _vp]),
    'estts_attention': (_i32, [C.POINTER(AttentionArgs), _vp]),
    'estts_rownorm': (_i32, [C.POINTER(RownormArgs), _vp]),
    'estts_embed_posenc': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    'estts_durations_cumsum': (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    'estts_expand': (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    'estts_conv2d_c1': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'estts_row_zero_mask': (_i32, [_vp, _i32, _vp, _i32, _i32, _vp]),
    'estts_cross_attention_row': (_i32, [C.POINTER(CrossRowArgs), _vp]),
    'estts_decoder_block_row': (_i32, [C.POINTER(DecoderBlockArgs), _vp]),
    'estts_prenet_row': (_i32, [C.POINTER(PrenetRowArgs), _vp]),
    'estts_masked_mae': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp]),
    'estts_wavernn_prep': (_i32, [C.POINTER(WavernnPrepArgs), _vp]),
    'estts_gru_gates': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    'estts_wavernn_sample': (_i32, [C.POINTER(WavernnSampleArgs), _vp]),
    'estts_wavernn_unfold': (_i32, [C.POINTER(WavernnUnfoldArgs), _vp]),
    'estts_wavernn_layer': (_i32, [C.POINTER(WavernnLayerArgs), _vp]),
    'estts_wavernn_cond': (_i32, [C.POINTER(WavernnCondArgs), _vp]),
    'estts_pack_weight': (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    'estts_gemm_wgrad': (_i32, [C.POINTER(WgradArgs), _vp]),
    'estts_attention_bwd': (_i32, [C.POINTER(AttentionBwdArgs), _vp]),
    'estts_rownorm_bwd': (_i32, [C.POINTER(RownormBwdArgs), _vp]),
    'estts_colreduce': (_i32, [C.POINTER(ColreduceArgs), _vp]),
    'estts_bn_finalize': (_i32, [_vp, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'estts_affine_act': (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    'estts_bn_bwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    'estts_gru_fused_fwd': (_i32, [C.POINTER(GruFusedArgs), _vp]),
    'estts_gru_fused_bwd': (_i32, [C.POINTER(GruFusedArgs), _vp]),
    'estts_gru_fused_fwd2': (_i32, [C.POINTER(GruFusedArgs), C.POINTER(GruFusedArgs), _vp]),
    'estts_gru_fused_bwd2': (_i32, [C.POINTER(GruFusedArgs), C.POINTER(GruFusedArgs), _vp]),
    'estts_bn_finalize_slots': (_i32, [_vp, _i64, _vp, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'estts_bn_bwd_slots': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    'estts_dropout': (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _f32, _u32, _u32, _vp]),
    'estts_axpby': (_i32, [_vp, _i32, _f32, _vp, _i32, _f32, _i32, _i32, _vp]),
    'estts_head_scale': (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    'estts_mine_divergence': (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _f32, _vp]),
    'estts_mse': (_i32, [_vp, _vp, _i32, _vp, _vp, _f32, _vp]),
    'estts_club_terms': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _f32, _vp]),
    'estts_im2col': (_i32, [_vp, _vp, _i32] + [_i32] * 11 + [_vp]),
    'estts_col2im': (_i32, [_vp, _i32, _vp] + [_i32] * 11 + [_vp]),
    'estts_embed_bwd': (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _vp]),
    'estts_expand_bwd': (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    'estts_masked_mae_bwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _vp]),
    'estts_adam': (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp]),
    'estts_gru_step_fwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    'estts_gru_step_bwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    'estts_wavernn_prep_bwd': (_i32, [C.POINTER(WavernnPrepBwdArgs), _vp]),
    'estts_softmax_xent': (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    'estts_scaled_xent': (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
    'estts_gemm_f32': (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    'estts_spec_magnitude': (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp]),
    'estts_mel_normalize': (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp]),
    'estts_alignment_lengths': (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    'estts_alignment_stats': (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'estts_alignment_peaks': (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    'estts_pack_weights': (_i32, [_vp, _i32, _i64, _vp]),
    'estts_run_program': (_i32, [C.POINTER(ProgOp), _i32, _i32, _vp]),
    'estts_mol_loss': (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
}

_lib = None


class EsttsError(RuntimeError):
    pass


def load():
    """Load libestts.so once; raise if it is missing, stale or lacks a declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EsttsError(f'{LIB_PATH} not found: the HIP extension is not built (make -C {_PKG}/csrc). '
                         'There is no CPU fallback for the hot path.')
    # The library's code objects register with the HIP runtime torch ships.  That runtime has to be initialised by torch
    # first: registered into an uninitialised runtime, every later launch reports "no ROCm-capable device is detected".
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EsttsError(f'{LIB_PATH} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    v = lib.estts_abi_version()
    if v != ABI_VERSION:
        raise EsttsError(f'libestts.so ABI version {v}, host code expects {ABI_VERSION}: rebuild')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().estts_last_error()
        raise EsttsError(f'{what} failed (rc={rc}): {msg.decode() if msg else "?"}')
