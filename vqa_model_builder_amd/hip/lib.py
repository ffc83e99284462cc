"""ctypes binding of libvqa_hip.so (include/vqa_hip.h).  Fails loudly when the HIP library is missing:
there is no CPU or eager-PyTorch fallback anywhere on the product path."""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'csrc', 'libvqa_hip.so')              # bfloat16 operands
LIB_PATH_F16 = os.path.join(os.path.dirname(_HERE), 'csrc', 'libvqa_hip_f16.so')      # IEEE fp16 operands (same sources, -DVQA_HALF_F16)

vp, i32, f32, u64, u32, sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_size_t


class VqaGemmDesc(C.Structure):
    _fields_ = [('a', vp), ('b', vp), ('M', i32), ('N', i32), ('K', i32), ('lda', i32), ('ldb', i32),
                ('a_kc', i32), ('b_kc', i32),
                ('c_f32', vp), ('ldc_f32', i32), ('c_bf16', vp), ('ldc_bf16', i32),
                ('pre_bf16', vp), ('ld_pre', i32), ('bias', vp), ('residual', vp), ('ld_res', i32),
                ('act_grad_of', vp), ('ld_ag', i32), ('act', i32), ('act_bwd', i32), ('alpha', f32),
                ('drop_p', f32), ('drop_seed', u64), ('drop_stream', u32),
                ('split_k', i32), ('allow_split_k', i32), ('tile_hint', i32), ('colsum', vp), ('c_prezeroed', i32)]


class VqaAttnDesc(C.Structure):
    _fields_ = [('q', vp), ('k', vp), ('v', vp), ('o', vp), ('ldq', i32), ('ldk', i32), ('ldv', i32), ('ldo', i32),
                ('B', i32), ('H', i32), ('Sq', i32), ('Skv', i32), ('Dh', i32), ('key_padding_mask', vp),
                ('scale', f32), ('drop_p', f32), ('drop_seed', u64), ('drop_stream', u32),
                ('d_o', vp), ('ldd_o', i32), ('dq', vp), ('dk', vp), ('dv', vp),
                ('lddq', i32), ('lddk', i32), ('lddv', i32), ('dq_colsum', vp), ('dk_colsum', vp), ('dv_colsum', vp), ('ws', vp), ('causal', i32)]


class VqaFusedAttnDesc(C.Structure):
    _fields_ = [('xq', vp), ('ldxq', i32), ('xkv', vp), ('ldxkv', i32), ('w_in', vp), ('ldw', i32), ('b_in', vp),
                ('q', vp), ('k', vp), ('v', vp), ('ldq', i32), ('ldk', i32), ('ldv', i32), ('o', vp), ('ldo', i32),
                ('B', i32), ('H', i32), ('Sq', i32), ('Skv', i32), ('D', i32), ('key_padding_mask', vp),
                ('scale', f32), ('drop_p', f32), ('drop_seed', u64), ('drop_stream', u32), ('causal', i32)]


class VqaGemmGroupItem(C.Structure):
    _fields_ = [('a', vp), ('b', vp), ('c_f32', vp), ('M', i32), ('N', i32), ('K', i32), ('lda', i32), ('ldb', i32), ('ldc', i32)]


class VqaLnReduceItem(C.Structure):
    _fields_ = [('ws', vp), ('nblocks', i32), ('cols', i32), ('dgamma', vp), ('dbeta', vp), ('dx_colsum', vp)]


class VqaAdamWDesc(C.Structure):
    _fields_ = [('param', vp), ('grad', vp), ('exp_avg', vp), ('exp_avg_sq', vp), ('param_bf16', vp), ('n', u64),
                ('lr', f32), ('beta1', f32), ('beta2', f32), ('eps', f32), ('weight_decay', f32),
                ('bias_correction1', f32), ('bias_correction2', f32), ('grad_scale', vp)]


# name -> (restype, argtypes); must list every symbol include/vqa_hip.h declares (tests check this)
SIGNATURES = {
    'vqa_abi_version': (i32, []),
    'vqa_half_kind': (i32, []),
    'vqa_gemm_bf16': (i32, [C.POINTER(VqaGemmDesc), vp]),
    'vqa_gemm_profile': (None, [i32, i32]),
    'vqa_gemm_profile_collect': (i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'vqa_gemm_profile_collect2': (i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    'vqa_set_gemm_ws': (None, [i32]),
    'vqa_set_gemm_use_tr': (None, [i32]),
    'vqa_set_gemm_v1_fast': (None, [i32]),
    'vqa_set_gemm_pipeline': (None, [i32]),
    'vqa_cast_f32_bf16': (i32, [vp, vp, sz, vp]),
    'vqa_cast_multi': (i32, [vp, i32, u64, vp]),
    'vqa_cast_bf16_f32': (i32, [vp, vp, sz, vp]),
    'vqa_colsum_bf16': (i32, [vp, i32, i32, i32, vp, vp]),
    'vqa_colsum_f32': (i32, [vp, i32, i32, i32, vp, vp]),
    'vqa_add_f32': (i32, [vp, vp, vp, vp, sz, vp]),
    'vqa_prefetch': (i32, [vp, sz, i32, vp]),
    'vqa_act_drop_bwd': (i32, [vp, vp, i32, vp, vp, sz, f32, u64, u32, vp]),
    'vqa_gather_rows_f32': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_patchify_bf16': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'vqa_clip_assemble': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_clip_assemble_bwd': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_layernorm_fwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, u64, u32, vp]),
    'vqa_layernorm_bwd_ws_floats': (sz, [i32]),
    'vqa_layernorm_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, u64, u32, i32, vp]),
    'vqa_set_attention_mfma': (None, [i32]),
    'vqa_attention_fwd': (i32, [C.POINTER(VqaAttnDesc), vp]),
    'vqa_fused_inproj_attention_fwd': (i32, [C.POINTER(VqaFusedAttnDesc), vp]),
    'vqa_attention_bwd': (i32, [C.POINTER(VqaAttnDesc), vp]),
    'vqa_roberta_embed_fwd': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    'vqa_roberta_embed_bwd': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'vqa_embedding_rows_bwd': (i32, [vp, vp, vp, i32, i32, i32, vp]),
    'vqa_softmax_ce_argmax_fwd': (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, f32, vp]),
    'vqa_softmax_ce_bwd': (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    'vqa_router_gate_fwd': (i32, [vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_router_gate_bwd': (i32, [vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_router_topk_fwd': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_router_topk_bwd': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_router_aux_loss': (i32, [vp, vp, i32, i32, i32, f32, vp, vp]),
    'vqa_moe_expert_tokens': (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    'vqa_moe_scatter_add': (i32, [vp, vp, vp, vp, i32, i32, vp]),
    'vqa_moe_combine_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    'vqa_moe_route_weight_grad': (i32, [vp, vp, vp, i32, i32, i32, vp]),
    'vqa_moe_dense_combine_fwd': (i32, [vp, vp, vp, i32, i32, i32, vp]),
    'vqa_moe_dense_combine_bwd': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_act_drop_fwd': (i32, [vp, vp, vp, vp, sz, i32, f32, u64, u32, vp]),
    'vqa_glu_fwd': (i32, [vp, vp, i32, i32, f32, u64, u32, vp]),
    'vqa_glu_bwd': (i32, [vp, vp, vp, i32, i32, f32, u64, u32, vp]),
    'vqa_rows_mask_cast': (i32, [vp, i32, vp, i32, vp, vp, i32, i32, f32, u64, u32, vp]),
    'vqa_head_keep_fwd': (i32, [vp, vp, i32, i32, i32, i32, f32, u64, u32, vp]),
    'vqa_head_keep_bwd': (i32, [vp, vp, i32, i32, i32, i32, f32, u64, u32, vp]),
    'vqa_repeat_rows_f32': (i32, [vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    'vqa_rows_mean_f32': (i32, [vp, i32, vp, vp, i32, i32, i32, vp]),
    'vqa_take_stride_bf16': (i32, [vp, vp, sz, i32, i32, vp]),
    'vqa_scatter_stride_f32': (i32, [vp, vp, sz, i32, i32, vp]),
    'vqa_randn_f32': (i32, [vp, u64, u64, u32, vp]),
    'vqa_dropout_f32': (i32, [vp, vp, vp, u64, f32, u64, u32, vp]),
    'vqa_gemm_bf16_grouped': (i32, [vp, i32, i32, i32, vp]),
    'vqa_gemm_bf16_grouped2': (i32, [vp, i32, i32, i32, vp, vp]),
    'vqa_set_gemm_dw256': (None, [i32]),
    'vqa_set_gemm_group_tile': (None, [i32]),
    'vqa_set_gemm_force': (None, [i32, i32]),
    'vqa_set_gemm_tile_order': (None, [i32]),
    'vqa_set_gemm_k_rotate': (None, [i32]),
    'vqa_layernorm_bwd_blocks': (i32, [i32]),
    'vqa_set_layernorm_bwd_blocks': (None, [i32]),
    'vqa_layernorm_bwd_partials': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, f32, u64, u32, i32, vp]),
    'vqa_layernorm_reduce_grouped': (i32, [vp, i32, vp]),
    'vqa_outer_bf16': (i32, [vp, vp, vp, i32, i32, i32, vp]),
    'vqa_outer_bwd': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_attention_bwd_ws_floats': (u64, [i32, i32, i32, i32, i32]),
    'vqa_opt_chunk_elems': (i32, []),
    'vqa_sumsq_multi': (i32, [vp, vp, i32, vp, vp]),
    'vqa_adamw_multi': (i32, [vp, vp, i32, vp, f32, f32, f32, f32, f32, f32, f32, vp, f32, vp, vp]),
    'vqa_amp_update': (i32, [vp, vp, f32, f32, i32, vp]),
    'vqa_opt_advance_counts': (i32, [vp, vp, i32, vp, vp]),
    'vqa_opt_advance': (i32, [vp, vp, vp]),
    'vqa_adamw_step': (i32, [C.POINTER(VqaAdamWDesc), vp]),
    'vqa_sumsq_f32': (i32, [vp, u64, vp, vp]),
}

_lib = None
_libs = {}            # half kind -> typed handle
_half = 'bf16'        # operand type of the ACTIVE library: 'bf16' (default) | 'fp16'


class HipLibraryMissing(RuntimeError):
    pass


def half() -> str:
    return _half


def set_half(kind: str):
    """Selects which build of the kernels every later launch uses: 'bf16' (torch autocast's bf16 scheme) or 'fp16' (what the
    reference's main loop runs under: autocast fp16 + GradScaler, training_pipeline.py:346-347,457).  Process-wide; switch
    BEFORE building / moving a model (weight shadows and saved activations are of the active type)."""
    global _half, _lib
    if kind not in ('bf16', 'fp16'):
        raise ValueError(f"compute dtype must be 'bf16' or 'fp16' (got {kind!r})")
    if kind != _half:
        _half, _lib = kind, None
    return load()


SUMSQ_SLOTS, SUMSQ_STRIDE = 64, 32      # include/vqa_hip.h: VQA_SUMSQ_SLOTS / VQA_SUMSQ_STRIDE (vqa_gemm_bf16_grouped2's partial accumulators)


def load(path: str = None):
    """Loads the shared library of the active operand type and types every entry point.  Raises HipLibraryMissing if it
    is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is None and _half in _libs:
        _lib = _libs[_half]
        return _lib
    # torch ships its own libamdhip64: it must be in the process BEFORE this library is dlopen-ed, so that both bind to
    # the same HIP runtime (otherwise torch's streams / allocations are foreign to our launches: hipErrorNoDevice)
    import torch  # noqa: F401
    # VQA_HIP_LIB: an alternative bf16 build of the same sources (A/B experiments: scratch/ab_build.sh); never set in production
    p = path or (LIB_PATH_F16 if _half == 'fp16' else os.environ.get('VQA_HIP_LIB', LIB_PATH))
    if not os.path.exists(p):
        raise HipLibraryMissing(
            f'{p} not found: build it with `python -m vqa_model_builder_amd.csrc.build` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.')
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = res, args
    if os.environ.get('VQA_DW256') is not None:               # A/B experiments only: 0 = weight gradients on the 128 x 128 ring kernel (round 2)
        lib.vqa_set_gemm_dw256(int(os.environ['VQA_DW256']))
    if os.environ.get('VQA_GEMM_FAST') is not None:           # A/B experiments only: 0 = general-form ring kernels everywhere
        lib.vqa_set_gemm_v1_fast(int(os.environ['VQA_GEMM_FAST']))
    if os.environ.get('VQA_GEMM_K_ROTATE') is not None:       # A/B experiments only (like VQA_HIP_LIB): per-XCD k rotation of the ring GEMMs off / on
        lib.vqa_set_gemm_k_rotate(int(os.environ['VQA_GEMM_K_ROTATE']))
    if path is None:
        if lib.vqa_half_kind() != (1 if _half == 'fp16' else 0):
            raise HipLibraryMissing(f'{p} was built for the other operand type: rebuild (python -m vqa_model_builder_amd.csrc.build --force)')
        _lib = _libs[_half] = lib
    return lib
