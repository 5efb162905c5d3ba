"""ctypes binding of libpyprob_amd.so (include/pyprob_amd.h). There is NO fallback: if the HIP library is missing or
cannot be loaded, importing the compute path raises -- the product never routes through a CPU implementation."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libpyprob_amd.so')

PP_ABI_VERSION = 14
PP_MAX_OBS = 8
PP_MAX_LSTM_DEPTH = 4
PP_MAX_OBS_DEPTH = 4
PP_ADDR_TABLE_COLS = 8
PP_HEAD_NORMAL_MIXTURE, PP_HEAD_TRUNCNORMAL_MIXTURE, PP_HEAD_CATEGORICAL, PP_HEAD_POISSON_TN_MIXTURE = 0, 1, 2, 3
PP_HEAD_BERNOULLI = 4
PP_LOSS_BACKWARD, PP_LOSS_ZERO_GRADS, PP_LOSS_KEEP_LP = 1, 2, 4
PP_ADAM_ZERO_GRADS = 1
PP_ADAM_SCRATCH = 1056          # int32 per tensor (include/pyprob_amd.h)
PP_ADAM_SEEN = 1027


def larc_scratch_floats(n_params, n_tensors):
    """PP_LARC_SCRATCH_FLOATS (include/pyprob_amd.h)."""
    return 2 * (int(n_params) // 1024) + int(n_tensors)


PP_IS_STATS_SCRATCH = 6144   # doubles (include/pyprob_amd.h)

i32, i64, f32p, i32p, vp = C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p


class pp_addr(C.Structure):
    _fields_ = [('kind', i32), ('n_out', i32), ('hid', i32), ('smp_in', i32), ('dtype_id', i32), ('_pad', i32),
                ('addr_emb', i64), ('dtype_emb', i64), ('smp_w', i64), ('smp_b', i64),
                ('w1', i64), ('b1', i64), ('w2', i64), ('b2', i64)]


class pp_net(C.Structure):
    _fields_ = [('n_obs', i32),
                ('obs_in', i32 * PP_MAX_OBS), ('obs_hid', i32 * PP_MAX_OBS), ('obs_out', i32 * PP_MAX_OBS),
                ('obs_w0', i64 * PP_MAX_OBS), ('obs_b0', i64 * PP_MAX_OBS), ('obs_w1', i64 * PP_MAX_OBS),
                ('obs_b1', i64 * PP_MAX_OBS),
                ('e_obs', i32), ('smp_dim', i32), ('addr_dim', i32), ('dtype_dim', i32),
                ('fin_w0', i64), ('fin_b0', i64), ('fin_w1', i64), ('fin_b1', i64),
                ('lstm_in', i32), ('lstm_dim', i32),
                ('w_ih', i64), ('w_hh', i64), ('b_ih', i64), ('b_hh', i64),
                ('n_addr', i32), ('n_dtype', i32),
                ('addrs', C.POINTER(pp_addr)), ('addr_table', vp), ('n_params', i64),
                ('lstm_depth', i32), ('_pad2', i32),
                ('lstm_w_ih', i64 * PP_MAX_LSTM_DEPTH), ('lstm_w_hh', i64 * PP_MAX_LSTM_DEPTH),
                ('lstm_b_ih', i64 * PP_MAX_LSTM_DEPTH), ('lstm_b_hh', i64 * PP_MAX_LSTM_DEPTH),
                ('obs_depth', i32 * PP_MAX_OBS),
                ('obs_w', (i64 * PP_MAX_OBS_DEPTH) * PP_MAX_OBS), ('obs_b', (i64 * PP_MAX_OBS_DEPTH) * PP_MAX_OBS)]


class pp_batch(C.Structure):
    _fields_ = [('n_traces', i32), ('n_rows', i32), ('t_max', i32), ('obs_width', i32),
                ('n_active', C.POINTER(i32)), ('row_off', C.POINTER(i32)), ('grp_off', C.POINTER(i32)),
                ('obs', vp), ('value', vp), ('prior', vp), ('addr', vp), ('prev_row', vp), ('grp_rows', vp),
                ('trace', vp), ('row_off_dev', vp), ('nxt_off', C.POINTER(i32)), ('nxt_rows', vp)]


class pp_lw_term(C.Structure):
    _fields_ = [('kind', i32), ('p0_stride', i32), ('p1_stride', i32), ('x_stride', i32),
                ('p0', vp), ('p1', vp), ('x', vp), ('scale', C.c_float)]


class pp_gemm_args(C.Structure):
    _fields_ = [('A', vp), ('lda', i64), ('a_idx', vp),
                ('B', vp), ('ldb', i64), ('b_idx', vp),
                ('C', vp), ('ldc', i64), ('c_idx', vp),
                ('M', i32), ('N', i32), ('K', i32), ('a_kmajor', i32), ('b_kmajor', i32),
                ('bias', vp), ('bias2', vp), ('mask', vp), ('ldmask', i64), ('relu', i32), ('accumulate', i32),
                ('colsum', vp), ('split_k', i32), ('_pad', i32)]


class pp_pack_info(C.Structure):
    _fields_ = [(n, i64) for n in ('n_traces', 'n_rows', 't_max', 'device_words', 'obs', 'value', 'prior', 'addr', 'prev_row',
                                   'grp_rows', 'trace', 'row_off_dev', 'nxt_rows', 'n_active', 'row_off', 'grp_off',
                                   'nxt_off', 'order', 'src_row')]


class pp_shard_columns(C.Structure):
    _fields_ = [(n, vp) for n in ('trace_len', 'row_off', 'obs', 'value', 'prior', 'addr', 'addr_remap')]


class pp_train_buffers(C.Structure):
    _fields_ = [('params', vp), ('grads', vp), ('exp_avg', vp), ('exp_avg_sq', vp),
                ('chunk_tensor', vp), ('tensor_step', vp), ('adam_scratch', vp),
                ('workspace', vp), ('workspace_bytes', C.c_size_t),
                ('staging', vp), ('device_batch', vp), ('slot_words', i64),
                ('loss_ring', vp), ('status_ring', vp), ('n_tensors', i32), ('n_slots', i32),
                ('dp_world', i32), ('dp_n_skip', i32), ('dp_skip_off', i64 * 4), ('dp_skip_cnt', i64 * 4)]


class pp_tensor_roles(C.Structure):
    _fields_ = [('off', vp), ('addr', vp), ('role', vp)]


# name -> (restype, argtypes); every symbol include/pyprob_amd.h declares
PROTOTYPES = {
    'pp_train_slot_words': (i64, [i32, i64, i32, i32, i32, i32]),
    'pp_train_sync': (C.c_int, []),
    'pp_train_resident': (C.c_int, [C.POINTER(pp_net), C.POINTER(pp_train_buffers), vp, vp, i32, vp, C.c_float, C.c_float,
                                    C.c_float, C.c_float, i32, vp]),
    'pp_train_steps': (C.c_int, [C.POINTER(pp_net), C.POINTER(pp_train_buffers), C.POINTER(pp_tensor_roles), vp, i32, vp, i32,
                                 vp, vp, i32, vp, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp, vp]),
    'pp_pack_indexed': (C.c_int, [vp, i32, vp, vp, i32, i32, i32, vp, i64, vp]),
    'pp_pack_words': (i64, [i32, i64, i32, i32, i32]),
    'pp_pack_ragged': (C.c_int, [vp, vp, vp, vp, i32, vp, i32, i32, i32, vp, i64, vp]),
    'pp_abi_version': (C.c_int, []),
    'pp_last_error': (C.c_char_p, []),
    'pp_device_count': (C.c_int, []),
    'pp_ic_workspace_bytes': (C.c_size_t, [C.POINTER(pp_net), i32, i32]),
    'pp_ic_loss': (C.c_int, [C.POINTER(pp_net), C.POINTER(pp_batch), vp, vp, vp, C.c_size_t, vp, vp, vp, i32, vp]),
    'pp_adam_step': (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp, vp, i32, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_float, i32, vp, vp]),
    'pp_sgd_step': (C.c_int, [vp, vp, vp, i64, vp, vp, i32, C.c_float, C.c_float, i32, C.c_float, C.c_float, i32, vp, vp]),
    'pp_larc_scale': (C.c_int, [vp, vp, i64, vp, vp, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                i32, vp, vp, vp]),
    'pp_is_workspace_bytes': (C.c_size_t, [C.POINTER(pp_net), i32]),
    'pp_is_init': (C.c_int, [C.POINTER(pp_net), vp, vp, vp, vp, C.c_size_t, vp]),
    'pp_is_step': (C.c_int, [C.POINTER(pp_net), vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, C.c_uint64,
                             C.c_uint64, vp, C.c_size_t, vp]),
    'pp_is_step_rows': (C.c_int, [C.POINTER(pp_net), vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, C.c_uint64,
                                  C.c_uint64, vp, C.c_size_t, vp]),
    'pp_is_step_fused_supported': (C.c_int, [C.POINTER(pp_net), i32, i32]),
    'pp_is_statement_rows': (C.c_int, [C.POINTER(pp_net), vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32,
                                       C.c_uint64, C.c_uint64, vp, C.c_size_t, vp]),
    'pp_prior_draw': (C.c_int, [i32, vp, i32, vp, i32, i32, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp]),
    'pp_is_first_statement_supported': (C.c_int, [C.POINTER(pp_net), i32]),
    'pp_is_first_statement': (C.c_int, [C.POINTER(pp_net), vp, vp, i32, vp, vp, vp, vp, C.c_size_t, vp]),
    'pp_is_step_net': (C.c_int, [C.POINTER(pp_net), vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, C.c_size_t, vp]),
    'pp_is_fused': (C.c_int, [C.POINTER(pp_net), i32, i32, vp, C.POINTER(pp_lw_term), C.POINTER(C.c_int32), i32, vp, vp, i32,
                              C.c_uint64, C.c_uint64, vp, vp, vp, C.c_size_t, vp]),
    'pp_logweight_accumulate': (C.c_int, [i32, vp, i32, vp, i32, vp, i32, C.c_float, vp, vp, i32, vp]),
    'pp_logweight_accumulate_rows': (C.c_int, [i32, vp, i32, vp, i32, vp, i32, C.c_float, vp, vp, i32, vp]),
    'pp_copy_rows': (C.c_int, [vp, i32, vp, vp, i32, vp]),
    'pp_partition_rows': (C.c_int, [vp, vp, i32, vp, vp, vp, vp, vp]),
    'pp_partition_rows_polled': (C.c_int, [vp, vp, i32, vp, vp, vp, i32, vp, vp]),
    'pp_logweight_terms': (C.c_int, [C.POINTER(pp_lw_term), i32, vp, i32, i32, vp]),
    'pp_axpy': (C.c_int, [C.c_float, vp, vp, i32, vp]),
    'pp_is_stats': (C.c_int, [vp, vp, i32, vp, vp, vp]),
    'pp_gemm_f32': (C.c_int, [C.POINTER(pp_gemm_args), vp]),
    'pp_gemm_f32_grouped': (C.c_int, [C.POINTER(pp_gemm_args), i32, vp]),
    'pp_colsum_f32': (C.c_int, [vp, i64, vp, i32, i32, vp, vp, vp]),
    'pp_lstm_input_gather': (C.c_int, [C.POINTER(pp_net), vp, vp, vp, vp, vp, vp, i32, vp, i64, vp]),
    'pp_lstm_cell_fwd': (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
    'pp_lstm_cell_bwd': (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    'pp_head_logprob': (C.c_int, [i32, vp, i64, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, vp]),
    'pp_debug_clock_probe': (C.c_int, [i32, vp, vp, vp]),
    'pp_dp_unique_id': (C.c_int, [C.c_char_p, vp]),
    'pp_dp_init': (C.c_int, [C.c_char_p, vp, i32, i32]),
    'pp_dp_world': (C.c_int, []),
    'pp_dp_destroy': (C.c_int, []),
    'pp_dp_allreduce': (C.c_int, [vp, vp, vp, i32, vp]),
    'pp_dp_reduce_grads': (C.c_int, [vp, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp]),
    'pp_dp_overlap': (C.c_int, [vp, vp, i32]),
    'pp_dp_overlap_stats': (C.c_int, [i32, vp]),
    'pp_debug_timeline': (C.c_int, [vp]),
    'pp_debug_wgtrace': (C.c_int, [vp, C.c_int32, C.c_int32]),
    'pp_debug_wgrad_plan': (C.c_int, [vp, vp, i32, vp, i32, vp]),
    'pp_prof_arm': (C.c_int, [i32, i32]),
    'pp_prof_stride': (C.c_int, [i32]),
    'pp_prof_collect': (C.c_int, [vp, i32, vp, vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the C-ABI library (building is __graft_entry__.build()'s / pyprob_amd.build.build()'s job)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be in the process BEFORE this library
    # is dlopen'ed so that both resolve to the same runtime -- otherwise two runtimes coexist and every launch on a
    # torch stream fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError('%s not found: build it with `python -m pyprob_amd.build` (hipcc, gfx950). '
                              'pyprob_amd has no CPU fallback.' % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pp_abi_version() != PP_ABI_VERSION:
        raise HipLibraryError('ABI version mismatch: library %d, binding %d' % (lib.pp_abi_version(), PP_ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().pp_last_error()
        raise RuntimeError('libpyprob_amd %s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream_ptr(device_index=None):
    """hipStream_t of torch's current stream on the current (or given) device. torch._C._cuda_getCurrentRawStream is the raw
    handle without building a torch.cuda.Stream object (~0.3 us instead of ~6 us: a replayed posterior call is ~70 us)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)
    return torch.cuda.current_stream().cuda_stream
