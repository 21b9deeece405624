#!/usr/bin/env python3
"""bench.py — decode-step throughput of the hot path on MI355X, Llama-3-8B GPTQ-int4 (Marlin format) TP=1.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).
  step      = one decode step of the whole batch through 32 decoder layers + lm_head:
              per layer RMSNorm -> int4 qkv GEMM -> RoPE -> reshape_and_cache -> paged_attention -> int4 o_proj GEMM
              -> RMSNorm(+residual) -> int4 gate_up GEMM -> SiLU*mul -> int4 down GEMM; then final norm, fp16 lm_head,
              greedy argmax. Every per-layer op is one of this repo's HIP kernels called through the C-ABI; only the
              embedding gather, the fp16 lm_head GEMM (hipBLASLt via torch.matmul: a plain library GEMM) and the argmax
              are torch ops. The step is captured in a HIP graph (the reference decodes under CUDA graphs too:
              vllm/worker/model_runner.py:910-1111).
  value     = decoded tokens / s over all ranks (each rank = an independent TP=1 replica: weak scaling).
  --gpus N  : run as N ranks of one node. Under `python -m torch.distributed.run` (RANK / WORLD_SIZE set) this process IS
              one rank; started directly with --gpus N > 1 it first spawns the N ranks as child processes (before any GPU
              call of its own) and relays rank 0's JSON line.
  --tp N    : ONE model sharded over the N ranks (Megatron TP, config 5: Llama-3-70B AWQ TP=8): column-parallel qkv /
              gate_up, row-parallel o / down, each followed by an RCCL all-reduce of [batch, hidden] fp16 INSIDE the
              captured step (linear.py:791-793 -> parallel_state.py:273-293); value = tokens/s of the whole job (strong).
  roofline  = the quantized-GEMM kernel class (what the metric names), timed with HIP events on the launch stream;
              rooflines = that class and paged attention.
  cpu_baseline = the CPU oracle (a port: dequant + fp32 matmul, scalar attention) on a bounded sample, rank 0 only.
Synthetic data: random token ids / activations, random-init int4 weights of the Llama-3-8B shapes.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TF = 2500.0  # dense fp16/bf16

LLAMA3_8B = dict(hidden=4096, inter=14336, heads=32, kv_heads=8, head=128, layers=32, vocab=128256, group=128)
# one TP=8 rank of Llama-3-70B (8 q heads + 1 kv head, inter 28672 / 8 per rank, 80 layers)
LLAMA3_70B_TP8_RANK = dict(hidden=8192, inter=3584, heads=8, kv_heads=1, head=128, layers=80, vocab=128256 // 8, group=128)
LLAMA3_70B = dict(hidden=8192, inter=28672, heads=64, kv_heads=8, head=128, layers=80, vocab=128256, group=128)


def tp_shard(cfg, tp):
    """Per-rank shapes of Megatron TP: q heads / kv heads / intermediate / vocab divided by tp, kv heads
    replicated once tp exceeds their number (config.py:396-404)."""
    assert cfg["heads"] % tp == 0 and cfg["inter"] % (tp * 128) == 0, "tp must divide the heads and 128-column tiles of inter"
    c = dict(cfg)
    c["heads"] = cfg["heads"] // tp
    c["kv_heads"] = max(1, cfg["kv_heads"] // tp)
    c["inter"] = cfg["inter"] // tp
    c["vocab"] = cfg["vocab"] // tp
    c["attn_hidden"] = cfg["hidden"]
    return c

# BASELINE.json configs[1..4]. "int4" is the headline (the metric is quoted on it); the others are selected with
# --config and print the same JSON line for their own workload.
VARIANTS = {
    "int4": dict(model=LLAMA3_8B, kv="auto",
                 metric="decode tokens/sec, Llama-3-8B GPTQ-int4 (Marlin-format) TP=1",
                 workload="Llama-3-8B GPTQ-int4 g128 decode step, TP=1 (configs[1])"),
    "sparse24": dict(model=LLAMA3_8B, kv="auto",
                     metric="decode tokens/sec, Llama-3-8B 2:4-sparse + int4 (sparse-Marlin) TP=1",
                     workload="Llama-3-8B 2:4-sparse int4 g128 decode step (gptq_marlin_24_gemm), TP=1 (configs[2])"),
    "fp8": dict(model=LLAMA3_8B, kv="fp8",
                metric="decode tokens/sec, Llama-3-8B fp8 weights + fp8 KV TP=1",
                workload="Llama-3-8B fp8 W8A8 (dynamic per-tensor activation quant + scaled_mm) + fp8-e4m3 KV decode step, "
                         "TP=1 (configs[3])"),
    "gptq-exllama": dict(model=LLAMA3_8B, kv="auto",
                         metric="decode tokens/sec, Llama-3-8B GPTQ-int4 through gptq_gemm (exllama format) TP=1",
                         workload="Llama-3-8B GPTQ-int4 g128 decode step through gptq_gemm after gptq_shuffle - the one quantized "
                                  "GEMM the reference itself builds for ROCm (CMakeLists.txt:149); same model as configs[1]"),
    "awq70b": dict(model=LLAMA3_70B, kv="auto",
                   metric="decode tokens/sec, Llama-3-70B AWQ-int4, tensor-parallel over RCCL (use with --tp N)",
                   workload="Llama-3-70B AWQ-int4 g128 decode step sharded by --tp (configs[4]: TP=8 over xGMI, 2 RCCL all-reduces "
                            "per layer inside the captured step)"),
    "awq70b-tp8rank": dict(model=LLAMA3_70B_TP8_RANK, kv="auto",
                           metric="decode tokens/sec of ONE TP=8 rank, Llama-3-70B AWQ-int4 (no all-reduce in the timed step)",
                           workload="Llama-3-70B AWQ-int4 g128, the per-rank shard of TP=8 (configs[4]): compute of one rank, "
                                    "the 2 all-reduces per layer are not part of this single-GPU line"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="sequences decoded per step and per GPU (SURVEY 8d sweeps 1..256; "
                    "the headline is the largest: serving throughput, and the regime the north star's MFMA target for the linear "
                    "layers refers to)")
    ap.add_argument("--ctx", type=int, default=1024, help="KV context length of every sequence")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--config", choices=sorted(VARIANTS), default="int4", help="which BASELINE.json config to run")
    ap.add_argument("--tp", type=int, default=0, help="tensor-parallel degree: all ranks share ONE model (needs --gpus == --tp); "
                    "0 = independent replicas")
    ap.add_argument("--no-fuse", action="store_true", help="int4: run the plain op sequence (split-K reduce launches, separate "
                    "rotary / reshape_and_cache) instead of the fused consumers - bit-identical results, more launches")
    ap.add_argument("--awq-op", action="store_true", help="AWQ configs: call the checkpoint-layout awq_gemm op instead of the "
                    "load-time repack + zero-point Marlin kernel that AWQLinearMethod uses")
    ap.add_argument("--attn", choices=["auto", "v1", "v2"], default="auto", help="decode attention op: auto = the reference's rule "
                    "(paged_attn.py:120-121)")
    ap.add_argument("--no-act-fuse", action="store_true", help="int4: gate_up GEMM and silu_and_mul as two ops (A/B of the fused epilogue)")
    ap.add_argument("--attn-fuse", action="store_true", help="int4: paged_attention_v2's partition launch, then o_proj with the reduce in "
                    "its prologue (needs NMX_GEMM_ATTN=rows in the environment; measured level with the reduce launch, off by default)")
    ap.add_argument("--no-norm-fuse", action="store_true", help="int4: fused_add_rms_norm and the GEMM behind it as two ops (A/B of the "
                    "norm-fused GEMM prologue at batch <= 4)")
    ap.add_argument("--no-attn-absmax", action="store_true", help="fp8: separate absmax pass over the attention output (A/B of paged_attention_v1/v2_absmax)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="also report batch 1/8/64/256 decode and the int4 GEMM TFLOP/s table")
    return ap.parse_args()


def random_marlin_weight(K, N, group, device, gen):
    """Random-init int4 weight already in Marlin layout: any int32 word is a valid packed tile row
    ([K/16, N*16/8]); scales [K/group, N] fp16 (Marlin-permuted order is irrelevant for random values)."""
    q = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=device, generator=gen)
    s = (torch.rand(K // group, N, device=device, generator=gen) * 0.004 + 0.002).to(torch.float16)
    return q, s


def random_weight(variant, K, N, group, device, gen):
    """Random-init weights of one linear layer in the layout the variant's op consumes."""
    if variant == "int4":
        return random_marlin_weight(K, N, group, device, gen)
    s = (torch.rand(K // group, N, device=device, generator=gen) * 0.004 + 0.002).to(torch.float16)
    if variant == "sparse24":
        # compressed non-zeros [K/2/16, N*16/8] + 2-bit positions [K/32, 2N] int16: every quad keeps elements 0 and 1
        # (nibble 0b0100) - a valid encoding; which positions are kept does not change the work
        q = torch.randint(-2**31, 2**31 - 1, (K // 32, N * 2), dtype=torch.int32, device=device, generator=gen)
        meta = torch.full((K // 32, N * 2), 0x4444, dtype=torch.int16, device=device)
        return q, meta, s
    if variant == "fp8":
        # [N, K] row-major fp8 weight, used as its column-major [K, N] transpose view (fp8.py:349-359)
        w = (torch.randn(N, K, device=device, generator=gen, dtype=torch.float16) * 0.02).to(torch.float8_e4m3fn)
        return w.t(), torch.full((1, ), 0.01, dtype=torch.float32, device=device)
    if variant == "gptq-exllama":
        q = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=device, generator=gen)
        z = torch.randint(-2**31, 2**31 - 1, (K // group, N // 8), dtype=torch.int32, device=device, generator=gen)
        return q, z, s, torch.empty(0, dtype=torch.int32, device=device)  # no act-order: empty g_idx (gptq.py:207-213)
    if variant in ("awq70b-tp8rank", "awq70b"):
        q = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device=device, generator=gen)
        z = torch.randint(-2**31, 2**31 - 1, (K // group, N // 8), dtype=torch.int32, device=device, generator=gen)
        return q, z, s
    raise ValueError(variant)


class Llama3Decode:
    """Synthetic Llama-3-8B decode step driver (the *caller* of the hot path; stands in for vllm's LlamaForCausalLM)."""

    def __init__(self, ops, cfg, batch, ctx, n_layers, device, block_size=16, variant="int4", all_reduce=None, all_gather=None,
                 awq_marlin=True, attn="auto"):
        self.ops, self.cfg, self.B, self.L, self.dev = ops, cfg, batch, ctx, device
        self.awq_marlin = awq_marlin  # AWQ configs: weights repacked at load (the layer's path); False = the raw awq_gemm op
        self.all_reduce, self.all_gather = all_reduce, all_gather  # tensor-parallel collectives (None: TP = 1)
        self.fuse = False  # int4 only: deferred split-K reduction + rotary / cache fusion (set by main)
        self.act_fuse = True  # int4: gate_up + silu_and_mul as one op (set by main)
        self.attn_absmax = True  # fp8: paged attention leaves the maxima of its output (set by main)
        self.attn_fuse = False  # int4: paged_attention_v2's reduce inside o_proj's prologue (off: level with the reduce launch; set by main)
        self.norm_fuse = True  # int4: fused_add_rms_norm + the GEMM behind it as one op (one launch at batch <= 4; set by main)
        self.n_layers = n_layers
        self.variant = variant
        self.kv_dtype = VARIANTS[variant]["kv"]
        H, I, nh, nkv, D = cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head"]
        self.q_size, self.kv_size = nh * D, nkv * D
        g = torch.Generator(device=device)
        g.manual_seed(0)
        self.shapes = dict(qkv=(H, self.q_size + 2 * self.kv_size), o=(self.q_size, H), gate_up=(H, 2 * I), down=(I, H))
        self.layers = []
        for _ in range(n_layers):
            lw = {}
            for name, (K, N) in self.shapes.items():
                lw[name] = random_weight(variant, K, N, cfg["group"], device, g)
                if variant == "gptq-exllama":
                    ops.gptq_shuffle(lw[name][0], lw[name][3], 4)  # exllama state machine, first apply (gptq.py:207-219)
                if variant in ("awq70b", "awq70b-tp8rank") and self.awq_marlin:
                    # AWQLinearMethod.process_weights_after_loading: one-time re-layout for the Marlin-format kernels
                    lw[name] = ops.awq_marlin_repack(lw[name][0], lw[name][1], lw[name][2])
            lw["ln1"] = torch.ones(H, dtype=torch.float16, device=device)
            lw["ln2"] = torch.ones(H, dtype=torch.float16, device=device)
            self.layers.append(lw)
        self.final_ln = torch.ones(H, dtype=torch.float16, device=device)
        self.lm_head = (torch.randn(cfg["vocab"], H, generator=g, device=device, dtype=torch.float16) * 0.02)
        self.embed = (torch.randn(cfg["vocab"], H, generator=g, device=device, dtype=torch.float16) * 0.02)
        # paged KV cache: every sequence owns ctx tokens; blocks scattered by a random permutation
        self.BS = block_size
        blocks_per_seq = (ctx + block_size - 1) // block_size
        NB = batch * blocks_per_seq
        scale = D**-0.5
        self.kv = []
        for _ in range(n_layers):
            if self.kv_dtype == "fp8":
                # fp8 caches are uint8 tensors with x = 16 (vllm/utils.py:471-472); bytes below 0x78 are finite e4m3 values
                kc = torch.randint(0, 0x78, (NB, nkv, D // 16, block_size, 16), dtype=torch.uint8, device=device, generator=g)
                vc = torch.randint(0, 0x78, (NB, nkv, D, block_size), dtype=torch.uint8, device=device, generator=g)
            else:
                kc = torch.empty(NB, nkv, D // 8, block_size, 8, dtype=torch.float16, device=device).uniform_(-scale, scale, generator=g)
                vc = torch.empty(NB, nkv, D, block_size, dtype=torch.float16, device=device).uniform_(-scale, scale, generator=g)
            self.kv.append((kc, vc))
        self.kv_scale = 0.05 if self.kv_dtype == "fp8" else 1.0
        self.block_tables = torch.randperm(NB, generator=g, device=device).to(torch.int32).reshape(batch, blocks_per_seq)
        self.seq_lens = torch.full((batch, ), ctx, dtype=torch.int32, device=device)
        last = ctx - 1  # the new token is written at position ctx-1 and attended with the ctx-1 cached ones
        self.slot_mapping = (self.block_tables[:, last // block_size].long() * block_size + last % block_size)
        self.positions = torch.full((batch, ), last, dtype=torch.long, device=device)
        inv_freq = 1.0 / (500000.0**(torch.arange(0, D, 2, device=device).float() / D))
        ang = torch.arange(ctx, device=device).float()[:, None] * inv_freq[None, :]
        self.cos_sin_cache = torch.cat((ang.cos(), ang.sin()), dim=-1).half()  # [max_pos, rot_dim] (rotary_embedding.py)
        self.tokens = torch.randint(0, cfg["vocab"], (batch, ), generator=g, device=device)
        # gptq_marlin: (N / 64) * 16 ints, gptq_marlin_24: (N / 128) * 64
        self.workspace = torch.zeros(max(N for _, N in self.shapes.values()) // 64 * 32, dtype=torch.int32, device=device)
        self.empty = torch.empty(0, dtype=torch.int32, device=device)
        self.scale = float(scale)
        # v2 temporaries (vllm/attention/ops/paged_attn.py:148-158)
        self.P = (ctx + 511) // 512
        self.use_v1 = ctx <= 8192 and (self.P == 1 or batch * nh > 512)  # paged_attn.py:120-121
        if attn != "auto":
            self.use_v1 = attn == "v1"
        if not self.use_v1:
            self.tmp_out = torch.empty(batch, nh, self.P, D, dtype=torch.float16, device=device)
            self.exp_sums = torch.empty(batch, nh, self.P, dtype=torch.float32, device=device)
            self.max_logits = torch.empty(batch, nh, self.P, dtype=torch.float32, device=device)
        self.next_tokens = torch.zeros(batch, dtype=torch.long, device=device)

    def gemm_as_in_step(self, x, w, name):
        """The GEMM launch the captured step issues for this layer: with fused consumers (the default) the deferred form - the
        GEMM kernel alone, its K-split slabs are summed by the consumer op -, otherwise the plain op (GEMM + reduce launch).
        kernel_breakdown() times this, so `roofline.avg_launch_us` is the duration of the kernel the roofline is about."""
        K, N = self.shapes[name]
        ops = self.ops
        if getattr(self, "fuse", False):
            if self.variant == "int4":
                return ops.gptq_marlin_gemm_deferred(x, w[0], w[1], self.empty, self.empty, self.workspace, 4, x.shape[0], N, K, True)
            if self.variant == "sparse24":
                return ops.gptq_marlin_24_gemm_deferred(x, w[0], w[1], w[2], self.workspace, 4, x.shape[0], N, K)
            if self.variant.startswith("awq70b") and self.awq_marlin:
                return ops.awq_marlin_gemm_deferred(x, w[0], w[1], w[2], x.shape[0], N, K)
        return self.gemm(x, w, name)

    def gemm(self, x, w, name):
        K, N = self.shapes[name]
        ops = self.ops
        if self.variant == "int4":
            return ops.gptq_marlin_gemm(x, w[0], w[1], self.empty, self.empty, self.workspace, 4, x.shape[0], N, K, True)
        if self.variant == "sparse24":
            return ops.gptq_marlin_24_gemm(x, w[0], w[1], w[2], self.workspace, 4, x.shape[0], N, K)
        if self.variant == "fp8":
            # Fp8LinearMethod.apply (fp8.py:340-359): dynamic per-tensor activation scale, then the scaled matmul
            qx, sx = ops.scaled_fp8_quant(x)
            return ops.cutlass_scaled_mm(qx, w[0], sx, w[1], torch.float16)
        if self.variant == "gptq-exllama":  # GPTQLinearMethod.apply (gptq.py:198-231); weights shuffled once at load
            return ops.gptq_gemm(x, w[0], w[1], w[2], w[3], True, 4)
        # AWQLinearMethod.apply (awq.py:166-172); after the load-time repack: the zero-point Marlin kernel
        if self.awq_marlin:
            return ops.awq_marlin_gemm(x, w[0], w[1], w[2], x.shape[0], N, K)
        return ops.awq_gemm(x, w[0], w[2], w[1], 8)

    def attention(self, q, layer, want_absmax=False):
        cfg = self.cfg
        kc, vc = self.kv[layer]
        out = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        if want_absmax:  # fp8 W8A8: the output's maxima as a by-product (o_proj's dynamic quantisation is then one launch)
            if self.use_v1:
                amax = self.ops.paged_attention_v1_absmax(out, q, kc, vc, cfg["kv_heads"], self.scale, self.block_tables,
                                                          self.seq_lens, self.BS, self.L, None, self.kv_dtype, self.kv_scale)
            else:
                amax = self.ops.paged_attention_v2_absmax(out, self.exp_sums, self.max_logits, self.tmp_out, q, kc, vc,
                                                          cfg["kv_heads"], self.scale, self.block_tables, self.seq_lens, self.BS,
                                                          self.L, None, self.kv_dtype, self.kv_scale)
            return out, amax
        if self.use_v1:
            self.ops.paged_attention_v1(out, q, kc, vc, cfg["kv_heads"], self.scale, self.block_tables, self.seq_lens,
                                        self.BS, self.L, None, self.kv_dtype, self.kv_scale)
        else:
            self.ops.paged_attention_v2(out, self.exp_sums, self.max_logits, self.tmp_out, q, kc, vc, cfg["kv_heads"],
                                        self.scale, self.block_tables, self.seq_lens, self.BS, self.L, None, self.kv_dtype,
                                        self.kv_scale)
        return out

    def step_fused(self):
        """The same layer arithmetic with 9 instead of ~14 dependent launches: the split-K partial sums of every int4 GEMM
        are folded into the element-wise op that consumes them (nmx_*_splitk), and rotary_embedding + reshape_and_cache
        are one launch. Bit-identical to step() (tests/test_fused_gpu.py)."""
        cfg, ops = self.cfg, self.ops
        nh, nkv, D = cfg["heads"], cfg["kv_heads"], cfg["head"]
        e, ws = self.empty, self.workspace

        def gemm(x, w, name):
            K, N = self.shapes[name]
            if self.variant == "sparse24":
                return ops.gptq_marlin_24_gemm_deferred(x, w[0], w[1], w[2], ws, 4, x.shape[0], N, K)
            if self.variant != "int4":  # AWQ repacked onto the Marlin kernel (awq70b configs)
                return ops.awq_marlin_gemm_deferred(x, w[0], w[1], w[2], x.shape[0], N, K)
            return ops.gptq_marlin_gemm_deferred(x, w[0], w[1], e, e, ws, 4, x.shape[0], N, K, True)

        h = self.embed[self.tokens]
        resid = h
        x = torch.empty_like(h)
        ops.rms_norm(x, h, self.layers[0]["ln1"], 1e-5)
        qkv_next = None
        for li, lw in enumerate(self.layers):
            kc, vc = self.kv[li]
            qkv_g = qkv_next if qkv_next is not None else gemm(x, lw["qkv"], "qkv")
            qkv_next = None
            qkv = ops.rope_reshape_and_cache(self.positions, qkv_g, nh, nkv, D, self.cos_sin_cache, kc, vc,
                                             self.slot_mapping, self.kv_dtype, self.kv_scale)
            if self.variant == "int4" and self.attn_fuse and not self.use_v1:
                # v2's partition launch, then o_proj with the reduce in its prologue (one launch at batch <= 16, else reduce + GEMM)
                parts = ops.paged_attention_v2_partials(qkv[:, :self.q_size].view(-1, nh, D), kc, vc, nkv, self.scale, self.block_tables,
                                                        self.seq_lens, self.BS, self.L, None, self.kv_dtype, self.kv_scale)
                K, N = self.shapes["o"]
                w = lw["o"]
                o = ops.paged_attention_gptq_marlin_gemm(parts, w[0], w[1], e, e, ws, 4, self.B, N, K, True)
            else:
                a = self.attention(qkv[:, :self.q_size].view(-1, nh, D), li)
                o = gemm(a.view(-1, nh * D), lw["o"], "o")
            if self.all_reduce is not None:
                self.all_reduce(o.materialize())
            if self.variant == "int4" and self.act_fuse and self.norm_fuse:
                # post-attention norm + gate_up + activation: one op (one launch at batch <= 4, else norm consumer + GEMM)
                K, N = self.shapes["gate_up"]
                w = lw["gate_up"]
                act, resid = ops.fused_add_rms_norm_gptq_marlin_gemm(o, resid, lw["ln2"], 1e-5, w[0], w[1], e, e, ws, 4, o.out.shape[0],
                                                                     N, K, True, silu_and_mul=True)
                h = None
            else:
                h = ops.fused_add_rms_norm_splitk(o, resid, lw["ln2"], 1e-5)
            if h is None:
                pass
            elif self.variant == "int4" and self.act_fuse:  # the activation runs in the GEMM's epilogue where the launch has no K split
                K, N = self.shapes["gate_up"]
                w = lw["gate_up"]
                act = ops.gptq_marlin_gemm_silu_and_mul(h, w[0], w[1], e, e, ws, 4, h.shape[0], N, K, True)
            else:
                act = torch.empty(h.shape[0], cfg["inter"], dtype=h.dtype, device=h.device)
                ops.silu_and_mul_splitk(act, gemm(h, lw["gate_up"], "gate_up"))
            d = gemm(act, lw["down"], "down")
            if self.all_reduce is not None:
                self.all_reduce(d.materialize())
            nxt = self.layers[li + 1]["ln1"] if li + 1 < self.n_layers else self.final_ln
            if self.variant == "int4" and self.norm_fuse and li + 1 < self.n_layers:
                # the next layer's input norm + qkv projection: one op (one launch at batch <= 4)
                K, N = self.shapes["qkv"]
                w = self.layers[li + 1]["qkv"]
                qkv_next, resid = ops.fused_add_rms_norm_gptq_marlin_gemm(d, resid, nxt, 1e-5, w[0], w[1], e, e, ws, 4, d.out.shape[0],
                                                                          N, K, True)
            else:
                x = ops.fused_add_rms_norm_splitk(d, resid, nxt, 1e-5)
        logits = torch.matmul(x, self.lm_head.t())
        if self.all_gather is not None:
            logits = self.all_gather(logits)
        self.next_tokens.copy_(logits.argmax(-1))
        return self.next_tokens

    def step_fused_fp8(self):
        """fp8 config with fewer launches per layer: the norm / activation in front of every fp8 linear layer leaves the
        per-token |max| of its output, so the dynamic activation quantisation is one launch instead of absmax + quantise
        (scaled_fp8_quant_partials: same scale and codes); the fp8 GEMMs leave their K-split slabs AND their scale epilogue
        to the consumer (cutlass_scaled_mm_deferred + the *_splitk_scaled ops: no reduce launch); rotary + KV-cache write
        are one launch. Round 3: paged attention leaves the maxima of its output too (paged_attention_v1/v2_absmax), so no
        stand-alone absmax launch is left in the step. Bit-identical to step() (tests/test_fused_gpu.py)."""
        cfg, ops = self.cfg, self.ops
        nh, nkv, D = cfg["heads"], cfg["kv_heads"], cfg["head"]

        def mm(x, amax, w):  # quantise (one launch when the producer left the maxima), deferred fp8 GEMM
            qx, sx = ops.scaled_fp8_quant_partials(x, amax) if amax is not None else ops.scaled_fp8_quant(x)
            return ops.cutlass_scaled_mm_deferred(qx, w[0], sx, w[1], torch.float16)

        h = self.embed[self.tokens]
        resid = h
        x = torch.empty_like(h)
        amax = ops.rms_norm_absmax(x, h, self.layers[0]["ln1"], 1e-5)
        for li, lw in enumerate(self.layers):
            kc, vc = self.kv[li]
            qkv = ops.rope_reshape_and_cache(self.positions, mm(x, amax, lw["qkv"]), nh, nkv, D, self.cos_sin_cache, kc, vc,
                                             self.slot_mapping, self.kv_dtype, self.kv_scale)
            if self.attn_absmax:
                a, amax = self.attention(qkv[:, :self.q_size].view(-1, nh, D), li, want_absmax=True)
            else:  # A/B: the two-launch quantisation of o_proj's input (absmax pass + quantise)
                a, amax = self.attention(qkv[:, :self.q_size].view(-1, nh, D), li), None
            h, amax = ops.fused_add_rms_norm_splitk(mm(a.view(-1, nh * D), amax, lw["o"]), resid, lw["ln2"], 1e-5, want_absmax=True)
            gu = mm(h, amax, lw["gate_up"])
            act = torch.empty(h.shape[0], cfg["inter"], dtype=h.dtype, device=h.device)
            amax = ops.silu_and_mul_splitk(act, gu, want_absmax=True)
            nxt = self.layers[li + 1]["ln1"] if li + 1 < self.n_layers else self.final_ln
            x, amax = ops.fused_add_rms_norm_splitk(mm(act, amax, lw["down"]), resid, nxt, 1e-5, want_absmax=True)
        logits = torch.matmul(x, self.lm_head.t())
        self.next_tokens.copy_(logits.argmax(-1))
        return self.next_tokens

    def step(self):
        """Same op sequence as the reference's LlamaDecoderLayer (vllm/model_executor/models/llama.py:154-230):
        fused_add_rms_norm -> qkv -> rotary_embedding (in place) -> reshape_and_cache -> paged_attention -> o_proj ->
        fused_add_rms_norm -> gate_up -> silu_and_mul -> down."""
        if self.fuse:
            return self.step_fused_fp8() if self.variant == "fp8" else self.step_fused()
        cfg, ops = self.cfg, self.ops
        nh, nkv, D = cfg["heads"], cfg["kv_heads"], cfg["head"]
        h = self.embed[self.tokens]
        resid = None
        for li, lw in enumerate(self.layers):
            if resid is None:
                resid = h
                x = torch.empty_like(h)
                ops.rms_norm(x, h, lw["ln1"], 1e-5)
            else:
                ops.fused_add_rms_norm(h, resid, lw["ln1"], 1e-5)
                x = h
            qkv = self.gemm(x, lw["qkv"], "qkv")
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
            ops.rotary_embedding(self.positions, q, k, D, self.cos_sin_cache, True)
            kc, vc = self.kv[li]
            ops.reshape_and_cache(k.view(-1, nkv, D), v.view(-1, nkv, D), kc, vc, self.slot_mapping, self.kv_dtype,
                                  self.kv_scale)
            a = self.attention(q.view(-1, nh, D), li)
            h = self.gemm(a.view(-1, nh * D), lw["o"], "o")
            if self.all_reduce is not None:
                self.all_reduce(h)  # RowParallelLinear: partial products of the K shards (linear.py:791-793)
            ops.fused_add_rms_norm(h, resid, lw["ln2"], 1e-5)
            gu = self.gemm(h, lw["gate_up"], "gate_up")
            act = torch.empty(gu.shape[0], cfg["inter"], dtype=gu.dtype, device=gu.device)
            ops.silu_and_mul(act, gu)
            h = self.gemm(act, lw["down"], "down")
            if self.all_reduce is not None:
                self.all_reduce(h)
        ops.fused_add_rms_norm(h, resid, self.final_ln, 1e-5)
        logits = torch.matmul(h, self.lm_head.t())
        if self.all_gather is not None:
            logits = self.all_gather(logits)  # vocab-parallel lm_head (logits_processor.py: gather of the shards)
        self.next_tokens.copy_(logits.argmax(-1))
        return self.next_tokens


def gemm_bytes(M, K, N, group, variant="int4"):
    """Algorithmic bytes of one linear layer call."""
    act = 2 * M * K + 2 * M * N
    scales = (K // group) * N * 2
    if variant == "sparse24":
        return K * N // 4 + K * N // 8 + scales + act       # kept values + 2-bit positions
    if variant == "fp8":
        return K * N + M * K + 2 * M * N                      # fp8 weights, fp8 activations
    if variant in ("awq70b-tp8rank", "awq70b", "gptq-exllama"):
        return K * N // 2 + scales + (K // group) * N // 2 + act  # + packed zero points
    return K * N // 2 + scales + act


def time_events(fn, reps):
    """Average device time of fn() in ms: fn's launches are captured into a HIP graph (so that Python / ctypes launch
    overhead does not pollute kernels that last a few microseconds) and the graph replays are bracketed by HIP
    events on the launch stream."""
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps  # ms


def kernel_breakdown(model, reps=3):
    """Per kernel class: avg launch duration (HIP events on the launch stream, all layers' distinct buffers in turn),
    algorithmic bytes / flops per launch, achieved GB/s."""
    B, cfg = model.B, model.cfg
    out = {}
    nl = model.n_layers
    for name, (K, N) in model.shapes.items():
        x = torch.randn(B, K, dtype=torch.float16, device=model.dev)

        def run(name=name, x=x):
            for lw in model.layers:
                model.gemm_as_in_step(x, lw[name], name)

        run()
        ms = time_events(run, reps) / nl
        by = gemm_bytes(B, K, N, cfg["group"], model.variant)
        fl = 2.0 * B * K * N
        out[("int4_gemm_" if model.variant == "int4" else model.variant + "_gemm_") + name] = dict(ms=ms, bytes=by, flops=fl, gbs=by / ms / 1e6, tflops=fl / ms / 1e9, launches=nl)
    q = torch.randn(B, cfg["heads"], cfg["head"], dtype=torch.float16, device=model.dev) * 0.1

    def run_attn():
        for li in range(nl):
            model.attention(q, li)

    run_attn()
    ms = time_events(run_attn, reps) / nl
    kvb = 1 if model.kv_dtype == "fp8" else 2
    by = 2 * B * model.L * cfg["kv_heads"] * cfg["head"] * kvb + 2 * B * cfg["heads"] * cfg["head"] * 2
    fl = 4.0 * B * model.L * cfg["heads"] * cfg["head"]
    out["paged_attention_" + ("v1" if model.use_v1 else "v2")] = dict(ms=ms, bytes=by, flops=fl, gbs=by / ms / 1e6,
                                                                      tflops=fl / ms / 1e9, launches=nl)
    return out


def cpu_baseline(cfg, batch, ctx):
    """Times the CPU oracle (port) on whole decoder layers' hot-path ops at this batch; scaled to all layers."""
    # threads = the CPU share of this box (affinity mask), capped at 32
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(ncpu, 32))
    os.environ["OMP_NUM_THREADS"] = str(threads)  # read when liboracle (OpenMP) is first loaded, below
    import oracle
    from oracle import packing
    torch.manual_seed(0)
    H, I, nh, nkv, D = cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head"]
    shapes = [(H, (nh + 2 * nkv) * D), (nh * D, H), (H, 2 * I), (I, H)]
    cpu_batch = min(batch, 256)
    gemm_in = []
    for K, N in shapes:
        mq = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32)
        ms = (torch.rand(K // cfg["group"], N) * 0.01 + 0.005).half()
        gemm_in.append((torch.randn(cpu_batch, K, dtype=torch.float16), mq, ms, K, N))
    # attention on cpu_batch sequences
    BS = 16
    nb = cpu_batch * ((ctx + BS - 1) // BS)
    kc = torch.empty(nb, nkv, D // 8, BS, 8, dtype=torch.float16).uniform_(-0.1, 0.1)
    vc = torch.empty(nb, nkv, D, BS, dtype=torch.float16).uniform_(-0.1, 0.1)
    q = torch.empty(cpu_batch, nh, D, dtype=torch.float16).uniform_(-0.1, 0.1)
    bt = torch.randperm(nb).to(torch.int32).reshape(cpu_batch, -1)
    sl = torch.full((cpu_batch, ), ctx, dtype=torch.int32)
    out = torch.empty_like(q)

    def one_layer():
        t0 = time.perf_counter()
        for a, mq, ms, K, N in gemm_in:
            oracle.gptq_marlin_gemm(a, mq, ms, None, None, None, 4, cpu_batch, N, K, True)
        oracle.paged_attention_v1(out, q, kc, vc, nkv, D**-0.5, bt, sl, BS, ctx, None, "auto", 1.0)
        return time.perf_counter() - t0

    # bounded sample: whole layers of the same workload until ~12 s of CPU work are spent (at least 1, at most 16 layers)
    times = [one_layer()]
    while sum(times) < 12.0 and len(times) < 16:
        times.append(one_layer())
    per_layer = sum(times) / len(times)
    step_s = per_layer * cfg["layers"]
    return dict(value=cpu_batch / step_s, unit="tokens/s", cores=threads, kind="port",
                sample=f"CPU oracle (dequant + fp32 matmul, scalar attention; OpenMP, {threads} threads of {ncpu} visible) on "
                f"{len(times)} of {cfg['layers']} layers (4 int4 GEMMs + paged attention each), batch {cpu_batch} x ctx {ctx}, "
                f"scaled to {cfg['layers']} layers; sample took {sum(times):.1f} s")


def pmc_traffic(kernels, args):
    """HBM bytes per launch (FETCH_SIZE x2-corrected + WRITE_SIZE, separate rocprofv3 --pmc passes of this same command;
    tools/profile_round.sh) from the newest committed profiles/rNN_kernel_summary.json. PMC counters cannot be read
    from inside a normal run, so the figure is only attached when the profiled workload is the one being run."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_kernel_summary.json")))
    if not files:
        return None, None
    try:
        prof = json.load(open(files[-1]))
        meta = prof.get("_workload", {"batch": 256, "ctx": 1024})
        if meta.get("batch") != args.batch or meta.get("ctx") != args.ctx or args.config != meta.get("config", "int4"):
            return None, None
        # bytes of all the class's kernels (main kernel(s) + split-K reduce where one ran) per call of the op:
        # calls = launches of the main kernels
        present = [k for k in kernels if k in prof]
        if not present:
            return None, None
        tot = sum((prof[k]["fetch_bytes_per_launch_corrected"] + prof[k]["write_bytes_per_launch"]) * prof[k]["launches"]
                  for k in present)
        calls = sum(prof[k]["launches"] for k in present if "reduce" not in k)
        return int(tot / max(calls, 1)), os.path.relpath(files[-1], here)
    except (KeyError, ValueError, OSError):
        return None, None


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (this parent never touches a GPU -
    nothing here may exec or re-exec a process that has initialised one) and relay rank 0's JSON line."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if have < n:
        print(json.dumps({"error": f"--gpus {n} but only {have} GPU(s) visible", "n_gpus": have}), flush=True)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    tp = args.tp
    if world > 1 or tp > 0:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:  # --tp 1 on one GPU: a world of one, still through RCCL
            import socket
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sock.getsockname()[1]))
            sock.close()
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)  # "nccl" IS RCCL on ROCm
    if tp > 0:
        assert tp == world, f"--tp {tp} needs exactly {tp} ranks (--gpus {tp}); got WORLD_SIZE={world}"

    from neuralmagic_vllm_amd import _custom_ops as ops
    var = VARIANTS[args.config]
    cfg = dict(var["model"])
    if args.layers is None:
        args.layers = cfg["layers"]
    cfg["layers"] = args.layers
    all_reduce = all_gather = None
    if tp > 0:
        cfg = tp_shard(cfg, tp)
        # the collectives run on the current (compute / capture) stream, so they are part of the captured step like the
        # reference's pynccl path (device_communicators/pynccl.py:99-118). world 1 still calls into RCCL.
        all_reduce = lambda t: dist.all_reduce(t)

        def all_gather(t):
            out = torch.empty(tp, t.shape[0], t.shape[1], dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t.contiguous())
            return out.movedim(0, 1).reshape(t.shape[0], tp * t.shape[1])

        warm = torch.ones(8, device=dev, dtype=torch.float16)
        dist.all_reduce(warm)  # communicator set-up outside the capture
        torch.cuda.synchronize()
    model = Llama3Decode(ops, cfg, args.batch, args.ctx, args.layers, dev, variant=args.config, all_reduce=all_reduce,
                         all_gather=all_gather, awq_marlin=not args.awq_op, attn=args.attn)

    # AWQ (hidden 8192): one workgroup per token reading 2-4 fp32 slabs of a 32-KiB row is slower than the 512-workgroup
    # reduce launch until there are >= 128 rows (measured: batch 64 7.81 vs 7.71 ms, batch 256 12.61 vs 13.28 ms)
    awq_fusable = args.config.startswith("awq70b") and not args.awq_op and args.batch >= 128
    model.act_fuse = not args.no_act_fuse
    model.attn_absmax = not args.no_attn_absmax
    model.norm_fuse = not args.no_norm_fuse
    model.attn_fuse = args.attn_fuse
    model.fuse = (args.config in ("int4", "fp8", "sparse24") or awq_fusable) and not args.no_fuse and (args.config != "fp8" or tp == 0)
    model.step()  # eager once: allocates GEMM scratch outside capture
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            model.step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            model.step()
    run = graph.replay if graph is not None else model.step

    for _ in range(args.warmup):
        run()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    replicas = 1 if tp > 0 else world
    value = replicas * args.batch * args.steps / dt
    ar_stats = None
    if tp > 0:
        # the step's all-reduce on its own: 2 per layer of [batch, hidden] fp16, chained in one graph
        buf = torch.randn(args.batch, cfg["hidden"], device=dev, dtype=torch.float16)
        n_ar = 2 * args.layers

        def ar_chain():
            for _ in range(n_ar):
                dist.all_reduce(buf)

        ar_ms = time_events(ar_chain, 5) / n_ar
        nbytes = buf.numel() * 2
        ar_stats = dict(bytes=nbytes, per_step=n_ar, us=round(ar_ms * 1e3, 2),
                        algbw_GBps=round(nbytes / ar_ms / 1e6, 1),
                        # ring all-reduce: every rank sends (and receives) 2 (n - 1) / n of the message over its ring links
                        busbw_GBps_per_link=round(2.0 * (tp - 1) / tp * nbytes / ar_ms / 1e6, 1),
                        step_share_ms=round(ar_ms * n_ar, 3))

    result = None
    if rank == 0:
        kb = kernel_breakdown(model)
        per_step = {k: v["ms"] * v["launches"] for k, v in kb.items()}
        # kernel classes: the four int4 GEMM launches of a layer are one kernel (marlin_gemm_kernel [+ its split-K reduce]);
        # per-launch figures are the mean over the four shapes, which is what rocprofv3's per-kernel average reports too.
        gem = [v for k, v in kb.items() if "_gemm_" in k]
        # int4: the class is named after the kernel that serves most of its launches at this batch (rows > 64: the wide /
        # ring kernels of marlin_wide.hip, else the row-block kernel); every kernel of the class is listed for the PMC sums
        int4_names = ("marlin_wide_kernel", "marlin_gemm_kernel") if args.batch > 64 else ("marlin_gemm_kernel", "marlin_wide_kernel")
        gemm_kernels = {"int4": int4_names + ("marlin_dma_kernel", "marlin_decode_kernel"),
                        "sparse24": ("marlin_gemm_kernel", "splitk_reduce_kernel"),
                        "fp8": ("scaled_mm_kernel", ),
                        "gptq-exllama": ("gptq_gemm_kernel", "splitk_reduce_kernel"),
                        "awq70b": ("marlin_gemm_kernel", "awq_gemm_kernel", "splitk_reduce_kernel"),
                        "awq70b-tp8rank": ("marlin_gemm_kernel", "awq_gemm_kernel", "splitk_reduce_kernel")}[args.config]
        att = [v for k, v in kb.items() if k.startswith("paged_attention")][0]
        gem_ms = sum(v["ms"] for v in gem)
        classes = {
            gemm_kernels[0]: dict(ms=gem_ms / len(gem), bytes=sum(v["bytes"] for v in gem) / len(gem),
                                  flops=sum(v["flops"] for v in gem) / len(gem), step_ms=gem_ms * gem[0]["launches"],
                                  prof=gemm_kernels),
            "paged_attention_kernel": dict(ms=att["ms"], bytes=att["bytes"], flops=att["flops"],
                                           step_ms=att["ms"] * att["launches"], prof=("paged_attention_kernel", )),
        }
        def roof_of(name):
            d = classes[name]
            d["gbs"] = d["bytes"] / d["ms"] / 1e6
            d["tflops"] = d["flops"] / d["ms"] / 1e9
            mfma_peak = MFMA_F16_PEAK_TF * (2.0 if (args.config == "fp8" and name != "paged_attention_kernel") else 1.0)  # fp8: 5 PF
            # HBM-bound below the ridge (4*M flop/B vs ~312 flop/B machine balance)
            hbm_bound = d["flops"] / d["bytes"] < mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
            traffic, traffic_src = pmc_traffic(d["prof"], args)
            common = dict(kernel=name, traffic=traffic, traffic_source=traffic_src, avg_launch_us=round(d["ms"] * 1e3, 2),
                          step_share_ms=round(d["step_ms"], 3), kernels_in_launch=list(d["prof"]))
            if hbm_bound:
                return dict(bound="hbm", achieved=round(d["gbs"], 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(d["gbs"] / HBM_PEAK_GBS, 4), algorithmic_bytes_per_launch=int(d["bytes"]), **common)
            return dict(bound="mfma", achieved=round(d["tflops"], 1), peak=mfma_peak, unit="TFLOP/s",
                        frac=round(d["tflops"] / mfma_peak, 4), algorithmic_flops_per_launch=float(d["flops"]),
                        algorithmic_bytes_per_launch=int(d["bytes"]),
                        traffic_over_algorithmic=(round(traffic / d["bytes"], 3) if traffic else None), **common)

        rooflines = {name: roof_of(name) for name in classes}
        # `roofline` = the kernel class the metric names: BASELINE.json's metric is "decode tokens/sec + int4 GEMM TFLOPS vs
        # roofline", so the quantized-GEMM class is reported whatever its share of the step (at batch
        # 256 paged attention has the larger share and used to take this slot); `rooflines` carries both classes
        roof = rooflines[gemm_kernels[0]]
        result = {
            "metric": var["metric"],
            "value": round(value, 1),
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if tp > 0 else "weak",
            "vs_baseline": None,
            "dtype": "fp8" if args.config == "fp8" else "f16",
            "data": "synthetic",
            "config": {"workload": var["workload"], "batch_per_gpu": args.batch,
                       "context": args.ctx, "layers": args.layers,
                       "kv_cache": ("fp8-e4m3" if model.kv_dtype == "fp8" else "fp16") + " block 16", "hip_graph": graph is not None,
                       "fused_consumers": bool(model.fuse),
                       "parallelism": (f"tp{tp} (one model over {tp} ranks, RCCL all-reduce in the captured step)" if tp > 0
                                       else f"dp{world} (independent TP=1 replicas)")},
            "roofline": roof,
            "rooflines": rooflines,
            "kernels": {k: {"us": round(v["ms"] * 1e3, 2), "GBps": round(v["gbs"], 1), "TFLOPs": round(v["tflops"], 2),
                            "step_share_ms": round(per_step[k], 3)} for k, v in kb.items()},
            "hot_path_share_of_step": round(sum(per_step.values()) / ms_per_step, 3),
        }
        if tp > 0:
            result["tp"] = tp
            result["allreduce"] = ar_stats
            # the communicator as torch.distributed / RCCL see it: ranks in the default group that
            # carried the all-reduces above, the backend name and the RCCL version linked into torch
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = None
            result["rccl"] = {"nranks": dist.get_world_size(), "backend": dist.get_backend(), "version": ver,
                              "rank0_device": str(dev)}
        if args.sweep and args.config == "int4":
            result["sweep"] = sweep(ops, cfg, dev)
        if not args.no_cpu_baseline and args.config == "int4" and world == 1:  # headline workload, single-GPU run only
            result["cpu_baseline"] = cpu_baseline(cfg, args.batch, args.ctx)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def sweep(ops, cfg, dev):
    """int4 GEMM table (4 Llama-3-8B shapes x M) — TFLOP/s and GB/s per launch."""
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    H, I = cfg["hidden"], cfg["inter"]
    shapes = dict(qkv=(H, 6144), o=(4096, H), gate_up=(H, 2 * I), down=(I, H))
    e = torch.empty(0, dtype=torch.int32, device=dev)
    table = {}
    for name, (K, N) in shapes.items():
        ws = [random_marlin_weight(K, N, cfg["group"], dev, g) for _ in range(4)]
        wsp = torch.zeros(N // 64 * 16, dtype=torch.int32, device=dev)
        for M in (1, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):
            x = torch.randn(M, K, dtype=torch.float16, device=dev)

            def run():
                for w in ws:
                    ops.gptq_marlin_gemm(x, w[0], w[1], e, e, wsp, 4, M, N, K, True)

            run()
            ms = time_events(run, 5) / len(ws)
            table[f"{name}_M{M}"] = {"us": round(ms * 1e3, 2), "GBps": round(gemm_bytes(M, K, N, cfg["group"]) / ms / 1e6, 1),
                                     "TFLOPs": round(2.0 * M * K * N / ms / 1e9, 2)}
    return table


if __name__ == "__main__":
    main()
