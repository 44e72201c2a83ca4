"""ctypes binding of libinsenticap_hip.so (include/insenticap_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call returns a
non-zero status the product raises.  (The CPU oracle under oracle/ is test infrastructure
and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ISC_HIP_LIB') or os.path.join(_HERE, 'lib', 'libinsenticap_hip.so')

ISC_MAX_SEG = 4

c_f32p = C.c_void_p  # device pointers travel as integers


class Seg(C.Structure):
    _fields_ = [('A', C.c_void_p), ('W', C.c_void_p), ('lda', C.c_int32), ('ldw', C.c_int32),
                ('K', C.c_int32), ('_pad', C.c_int32), ('A_hi', C.c_void_p), ('A_lo', C.c_void_p)]


class LinearProblem(C.Structure):
    _fields_ = [('seg', Seg * ISC_MAX_SEG), ('nseg', C.c_int32), ('M', C.c_int32), ('N', C.c_int32),
                ('relu', C.c_int32), ('bias0', C.c_void_p), ('bias1', C.c_void_p), ('bias2', C.c_void_p),
                ('keep_mask', C.c_void_p), ('mask_scale', C.c_float), ('ldc', C.c_int32),
                ('C', C.c_void_p), ('C_pre', C.c_void_p), ('accumulate', C.c_int32), ('_pad', C.c_int32),
                ('splitk_ws', C.c_void_p), ('splitk_ws_floats', C.c_int64)]


class LstmProblem(C.Structure):
    _fields_ = [('seg', Seg * ISC_MAX_SEG), ('nseg', C.c_int32), ('M', C.c_int32), ('H', C.c_int32),
                ('_pad', C.c_int32), ('b_ih', C.c_void_p), ('b_hh', C.c_void_p),
                ('c_prev', C.c_void_p), ('h_out', C.c_void_p), ('c_out', C.c_void_p),
                ('gates_out', C.c_void_p), ('h_keep_mask', C.c_void_p), ('mask_scale', C.c_float),
                ('hdrop_out', C.c_void_p), ('h_hi', C.c_void_p), ('h_lo', C.c_void_p),
                ('pre', C.c_void_p), ('tab', C.c_void_p),
                ('tab_ids', C.c_void_p), ('tab_ids_stride', C.c_int64),
                ('splitk_ws', C.c_void_p), ('splitk_ws_floats', C.c_int64)]


class ScanProblem(C.Structure):
    _fields_ = [('P', C.c_void_p), ('V', C.c_void_p), ('q', C.c_void_p), ('q2', C.c_void_p),
                ('w', C.c_void_p), ('w_bias', C.c_void_p), ('R', C.c_int32), ('A', C.c_int32),
                ('D', C.c_int32), ('rows', C.c_int32), ('out', C.c_void_p), ('alpha_out', C.c_void_p),
                ('alpha_ld', C.c_int64), ('out_hi', C.c_void_p), ('out_lo', C.c_void_p),
                ('row_ids', C.c_void_p), ('row_ids_ld', C.c_int64)]


class ScanBwdProblem(C.Structure):
    _fields_ = [('P', C.c_void_p), ('V', C.c_void_p), ('q', C.c_void_p), ('q2', C.c_void_p),
                ('w', C.c_void_p), ('alpha', C.c_void_p), ('dout', C.c_void_p), ('alpha_ld', C.c_int64),
                ('R', C.c_int32), ('A', C.c_int32), ('D', C.c_int32), ('accumulate', C.c_int32),
                ('dP', C.c_void_p), ('dV', C.c_void_p), ('dq', C.c_void_p), ('dw_rows', C.c_void_p),
                ('de_out', C.c_void_p), ('rows', C.c_int32), ('_pad', C.c_int32)]


class ScanGateArgs(C.Structure):
    _fields_ = [('scan', ScanProblem * 2), ('G', C.c_void_p * 2), ('zh', C.c_void_p), ('b_gc', C.c_void_p),
                ('b_gs', C.c_void_p), ('w_gate', C.c_void_p), ('b_gate', C.c_void_p), ('f', C.c_void_p),
                ('f_hi', C.c_void_p), ('f_lo', C.c_void_p), ('beta', C.c_void_p), ('beta_ld', C.c_int64)]


def _f(names, ctype):
    return [(n, ctype) for n in names.split()]


class StepPlan(C.Structure):
    """isc_step_plan: field order mirrors include/insenticap_hip.h exactly (checked by the layout test)."""
    _fields_ = (_f('rows H E A W V R Mw', C.c_int32) +
                _f('Wih1 Whh1 Wih2 Whh2 b_ih2 b_hh2 W_h2att b_h2att w_alpha_c b_alpha_c W_h2word b_h2word '
                   'w_alpha_s b_alpha_s W_gh b_gh W_gc b_gc W_gs b_gs w_gate b_gate W_cls b_cls '
                   'pre1 tab att_p att_e words_p words_e label_w xt tok', C.c_void_p) +
                [('tok_stride', C.c_int64)] +
                _f('h1_prev c1_prev h2_prev c2_prev h1 c1 h2 c2 g1 g2 qa v qw s z f alpha_c alpha_s beta',
                   C.c_void_p) +
                _f('alpha_c_ld alpha_s_ld beta_ld', C.c_int64) +
                [('out_mask', C.c_void_p), ('out_scale', C.c_float), ('apply_logsoftmax', C.c_int32),
                 ('hdrop', C.c_void_p), ('logits', C.c_void_p), ('ld_logits', C.c_int64),
                 ('pmax', C.c_void_p), ('psum', C.c_void_p), ('pidx', C.c_void_p),
                 ('splitk_ws', C.c_void_p), ('splitk_ws_floats', C.c_int64)] +
                _f('h1_prev_hi h1_prev_lo h2_prev_hi h2_prev_lo h1_hi h1_lo h2_hi h2_lo '
                   'v_hi v_lo s_hi s_lo f_hi f_lo words_ids', C.c_void_p) + [('words_ids_ld', C.c_int64)] +
                _f('gate_Gc gate_Gs', C.c_void_p) + _f('pair_rows_c _pad2', C.c_int32))


class StepBwdPlan(C.Structure):
    """isc_step_bwd_plan."""
    _fields_ = (_f('rows H E A W R Mw first last pair_rows_c', C.c_int32) +
                _f('Wih1 Whh1 Wih2 Whh2 W_h2att w_alpha_c W_h2word w_alpha_s W_gh W_gc W_gs w_gate '
                   'att_p att_e words_p words_e label_w g1 c1_prev c1 g2 c2_prev c2 qa qw v s z '
                   'alpha_c alpha_s beta', C.c_void_p) +
                _f('alpha_c_ld alpha_s_ld beta_ld', C.c_int64) +
                _f('dhd dG1 dG2 dG1_sum d_feat dh1 dv ds dh2_rec dh1_rec dc1_in dc2_in dc1_out dc2_out '
                   'dqa dqw dz dP_att dV_att dP_w dV_w dwc_rows dws_rows dwg_rows dbg_rows splitk_ws',
                   C.c_void_p) + [('splitk_ws_floats', C.c_int64), ('de_c', C.c_void_p), ('de_s', C.c_void_p)])


class BeamMergeArgs(C.Structure):
    _fields_ = ([('n_img', C.c_int32), ('beam', C.c_int32), ('T', C.c_int32), ('t', C.c_int32), ('eos_id', C.c_int64)] +
                _f('top_val top_idx score_in score_out last_in last_out words_in words_out len_in len_out done gather live',
                   C.c_void_p))


class RowsExt(C.Structure):
    """isc_rows_ext: what the few-row callers fold into the decode step (csrc/rows.hip)."""
    _fields_ = [('src_row', C.c_void_p), ('stats_tile', C.c_int32), ('beam', C.c_int32), ('cand_val', C.c_void_p),
                ('cand_idx', C.c_void_p), ('last_word', C.c_void_p), ('pad_id', C.c_int64), ('sos_id', C.c_int64),
                ('unk_id', C.c_int64), ('mask_special', C.c_int32), ('decoding_constraint', C.c_int32),
                ('row_div', C.c_int32), ('_pad', C.c_int32), ('live_in', C.c_void_p), ('fin_prev', C.c_void_p),
                ('fin_unfinished_out', C.c_void_p)]


class BeamSelectArgs(C.Structure):
    _fields_ = ([('n_img', C.c_int32), ('beam', C.c_int32), ('T', C.c_int32), ('t', C.c_int32), ('n_tile', C.c_int32),
                 ('V', C.c_int32), ('eos_id', C.c_int64)] +
                _f('part_max part_sum cand_val cand_idx score_in score_out last_in last_out words_in words_out len_in '
                   'len_out done src_row live top_val top_idx live_in state_in state_out', C.c_void_p) +
                [('state_planes', C.c_int32), ('H', C.c_int32)])


ISC_COLSUM_MAX_JOBS, ISC_COLSUM_MAX_OUT = 24, 3


class ColsumJob(C.Structure):
    _fields_ = [('x', C.c_void_p), ('ld', C.c_int64), ('M', C.c_int32), ('N', C.c_int32),
                ('out', C.c_void_p * ISC_COLSUM_MAX_OUT), ('n_out', C.c_int32), ('accumulate', C.c_int32)]


class RolloutStep(C.Structure):
    _fields_ = [('B', C.c_int32), ('V', C.c_int32), ('T', C.c_int32), ('t', C.c_int32),
                ('n_tile', C.c_int32), ('W', C.c_int32),
                ('part_max', C.c_void_p), ('part_sum', C.c_void_p), ('part_idx', C.c_void_p),
                ('logits', C.c_void_p), ('ld_logits', C.c_int64),
                ('forced', C.c_void_p), ('sample_u', C.c_void_p), ('eos_id', C.c_int64),
                ('seq', C.c_void_p), ('seq_logprobs', C.c_void_p), ('seq_masks', C.c_void_p),
                ('unfinished', C.c_void_p), ('alive', C.c_void_p), ('raw_tokens', C.c_void_p),
                ('emb', C.c_void_p), ('xt_add', C.c_void_p), ('xt_next', C.c_void_p)]


# name -> (restype, argtypes); every symbol include/insenticap_hip.h declares
SIGNATURES = {
    'isc_abi_version': (C.c_int, []),
    'isc_target_arch': (C.c_char_p, []),
    'isc_set_tile_override': (C.c_int, [C.c_int]),
    'isc_set_h3_mode': (C.c_int, [C.c_int]),
    'isc_h3_launches': (C.c_longlong, []),
    'isc_set_gemv_rows': (C.c_int, [C.c_int]),
    'isc_colsum_multi': (C.c_int, [C.POINTER(ColsumJob), C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    'isc_gemv_launches': (C.c_longlong, []),
    'isc_h3x_launches': (C.c_longlong, []),
    'isc_h3s_launches': (C.c_longlong, []),
    'isc_h3_weights_begin': (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    'isc_h3_weights_end': (C.c_int, [C.c_void_p]),
    'isc_h3_weights_suspend': (C.c_int, [C.c_void_p]),
    'isc_h3_weights_resume': (C.c_int, [C.c_void_p, C.c_void_p]),
    'isc_h3_weights_refresh': (C.c_int, [C.c_void_p]),
    'isc_linear_fwd': (C.c_int, [C.POINTER(LinearProblem), C.c_int, C.c_void_p]),
    'isc_gemm_bwd': (C.c_int, [C.POINTER(LinearProblem), C.c_int, C.c_int, C.c_void_p]),
    'isc_lstm_fwd': (C.c_int, [C.POINTER(LstmProblem), C.c_void_p]),
    'isc_step_fwd': (C.c_int, [C.POINTER(StepPlan), C.c_void_p]),
    'isc_set_stream_gate': (C.c_int, [C.c_void_p, C.c_void_p]),
    'isc_step_bwd': (C.c_int, [C.POINTER(StepBwdPlan), C.c_void_p]),
    'isc_rows_stats_tile': (C.c_int, [C.c_int]),
    'isc_rows_step_supported': (C.c_int, [C.POINTER(StepPlan)]),
    'isc_rows_step_fwd': (C.c_int, [C.POINTER(StepPlan), C.POINTER(RowsExt), C.c_void_p]),
    'isc_rows_vocab_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(RowsExt),
                                     C.c_void_p]),
    'isc_set_rows_nt': (C.c_int, [C.c_int]),
    'isc_set_rows_scan_max': (C.c_int, [C.c_int]),
    'isc_set_rows_scan_regions': (C.c_int, [C.c_int]),
    'isc_set_h3v': (C.c_int, [C.c_int]),
    'isc_set_h3_ksplit': (C.c_int, [C.c_int]),
    'isc_copy_multi': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    'isc_rows_launches': (C.c_longlong, []),
    'isc_beam_select': (C.c_int, [C.POINTER(BeamSelectArgs), C.c_void_p]),
    'isc_vocab_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'isc_sched_sample': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p]),
    'isc_sched_sample_raw': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p]),
    'isc_logsoftmax_apply': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    'isc_logsoftmax_apply_steps': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'isc_attn_scan_fwd': (C.c_int, [C.POINTER(ScanProblem), C.c_int, C.c_int, C.c_void_p]),
    'isc_attn_scan_gate_fwd': (C.c_int, [C.POINTER(ScanGateArgs), C.c_int, C.c_void_p]),
    'isc_gate_mix_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    'isc_embed_relu_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p]),
    'isc_embed_relu_mean_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p]),
    'isc_embed_senti_words_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64,
                                            C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    'isc_rollout_finalize': (C.c_int, [C.POINTER(RolloutStep), C.c_void_p]),
    'isc_rollout_finalize_launches': (C.c_longlong, []),
    'isc_beam_merge': (C.c_int, [C.POINTER(BeamMergeArgs), C.c_void_p]),
    'isc_beam_gather': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'isc_beam_topk': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    'isc_xe_loss_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p]),
    'isc_logsoftmax_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int64, C.c_int, C.c_void_p]),
    'isc_lstm_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'isc_attn_scan_bwd': (C.c_int, [C.POINTER(ScanBwdProblem), C.c_int, C.c_int, C.c_void_p]),
    'isc_attn_dv_from_alpha': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'isc_attn_dp_from_de': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'isc_gate_mix_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'isc_embed_relu_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                     C.c_int, C.c_int64, C.c_void_p, C.c_float, C.c_void_p, C.c_float,
                                     C.c_void_p, C.c_int64, C.c_void_p]),
    'isc_embed_relu_bwd_ws': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                        C.c_int, C.c_int64, C.c_void_p, C.c_float, C.c_void_p, C.c_float,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    'isc_colsum': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                             C.c_void_p]),
    'isc_relu_mask_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p,
                                    C.c_void_p]),
    'isc_xe_loss_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    'isc_xe_loss_bwd_sparse': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'isc_reward_loss_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'isc_reward_loss_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    'isc_logsoftmax_bwd_sparse': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    'isc_gather_logp_raw': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'isc_xe_loss_tokens_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'isc_logsoftmax_bwd_raw': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'isc_grad_scale': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_void_p, C.c_void_p]),
    'isc_splitk_workspace_bytes': (C.c_int64, [C.c_int64, C.c_int64]),
    'isc_h3_weights_workspace_bytes': (C.c_int64, [C.c_int64, C.c_int]),
    'isc_status': (C.c_int, [C.c_int]),
    'isc_set_status_words': (C.c_int, [C.c_void_p]),
    'isc_clamp_adam': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_double,
                                 C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                 C.c_void_p]),
    'isc_clamp_adam_hyper': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_void_p,
                                       C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
}

_ERRORS = {-1: 'ISC_E_NULL (required pointer is null)', -2: 'ISC_E_SHAPE (unsupported size)',
           -3: 'ISC_E_ALIGN (pointer / leading dimension not 16-byte aligned)',
           -4: 'ISC_E_WORKSPACE', -5: 'ISC_E_STATE'}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and type every entry point. Raises HipLibraryError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            'libinsenticap_hip.so not found at %s - build it with '
            '`python -m insenticap_model_amd._build` (hipcc, gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => header/library mismatch, surfaced loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = _ERRORS.get(rc, 'hipError_t %d' % rc if rc > 0 else 'error %d' % rc)
        raise HipLibraryError('%s failed: %s' % (what, msg))
