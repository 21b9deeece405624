"""AWQ — mirror of vllm/model_executor/layers/quantization/awq.py (config :13-73, method :76-176)."""
import os
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from neuralmagic_vllm_amd import _custom_ops as ops
from neuralmagic_vllm_amd.layers.quantization.base_config import LinearMethodBase, QuantizationConfig, set_weight_attrs


class AWQConfig(QuantizationConfig):

    def __init__(self, weight_bits: int, group_size: int, zero_point: bool) -> None:
        self.weight_bits, self.group_size, self.zero_point = weight_bits, group_size, zero_point
        if self.weight_bits != 4:
            raise ValueError("Currently, only 4-bit weight quantization is supported for "
                             f"AWQ, but got {self.weight_bits} bits.")
        self.pack_factor = 32 // self.weight_bits

    def __repr__(self) -> str:
        return f"AWQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, zero_point={self.zero_point})"

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half]

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        return cls(cls.get_from_keys(config, ["w_bit", "bits"]), cls.get_from_keys(config, ["q_group_size", "group_size"]),
                   cls.get_from_keys(config, ["zero_point"]))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["AWQLinearMethod"]:
        return AWQLinearMethod(self)

    def get_scaled_act_names(self) -> List[str]:
        return ["gelu", "gelu_fast", "gelu_new", "gelu_pytorch_tanh"]


class AWQLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: AWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        out_pp = sum(output_partition_sizes)
        if out_pp % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        packed = {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": cfg.pack_factor}
        qweight = Parameter(torch.empty(input_size_per_partition, out_pp // cfg.pack_factor, dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qweight, dict(packed))
        qzeros = Parameter(torch.empty(input_size_per_partition // cfg.group_size, out_pp // cfg.pack_factor, dtype=torch.int32),
                           requires_grad=False)
        set_weight_attrs(qzeros, dict(packed))
        scales = Parameter(torch.empty(input_size_per_partition // cfg.group_size, out_pp, dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {"input_dim": 0, "output_dim": 1})
        for name, prm in (("qweight", qweight), ("qzeros", qzeros), ("scales", scales)):
            layer.register_parameter(name, prm)
            set_weight_attrs(prm, extra_weight_attrs)

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        """MI355X-specific: re-lay the checkpoint tensors out for the Marlin-format kernels once (the per-call 8 x 8 nibble
        transposition of awq_gemm is what keeps that op at < 1 TB/s); the reference does the same kind of load-time
        repack for GPTQ -> Marlin (gptq_marlin.py:330-420). Falls back to awq_gemm where the shape does not allow it
        (group size not a multiple of 128, CPU tensors)."""
        qw = layer.qweight
        if not qw.is_cuda or getattr(layer, "marlin_q", None) is not None:
            return
        k, n = qw.shape[0], qw.shape[1] * self.quant_config.pack_factor
        if not ops.awq_marlin_supported(n, k, layer.scales.shape[0]) or layer.scales.dtype != torch.float16:
            return
        layer.marlin_q, layer.marlin_s, layer.marlin_z = ops.awq_marlin_repack(qw.data, layer.qzeros.data, layer.scales.data)
        layer.marlin_shape = (k, n)
        # The repacked layer never reads the checkpoint-layout tensors again (apply() takes the Marlin path for fp16 activations,
        # which is the only dtype the repack is done for): release them instead of keeping both layouts resident (+4.4 GB per
        # rank for Llama-3-70B at TP = 8, +5.7 GB for an 8B model at TP = 1 - KV-cache capacity). NMX_AWQ_KEEP_CHECKPOINT_LAYOUT=1
        # keeps them (A/B of the two device paths, bf16 activations on a repacked layer).
        if os.environ.get("NMX_AWQ_KEEP_CHECKPOINT_LAYOUT", "0") != "1":
            for name in ("qweight", "qzeros", "scales"):
                getattr(layer, name).data = torch.empty(0, dtype=getattr(layer, name).dtype, device=qw.device)

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        pack_factor = self.quant_config.pack_factor
        repacked = getattr(layer, "marlin_q", None) is not None
        out_shape = x.shape[:-1] + ((layer.marlin_shape[1] if repacked else layer.qweight.shape[-1] * pack_factor), )
        reshaped_x = x.reshape(-1, x.shape[-1])
        if repacked and reshaped_x.dtype != torch.float16 and layer.qweight.numel() == 0:
            raise RuntimeError("AWQ layer repacked for fp16 activations was called with " + str(reshaped_x.dtype) +
                               " (set NMX_AWQ_KEEP_CHECKPOINT_LAYOUT=1 to keep the checkpoint-layout tensors for other dtypes)")
        if repacked and reshaped_x.dtype == torch.float16:
            k, n = layer.marlin_shape
            out = ops.awq_marlin_gemm(reshaped_x.contiguous(), layer.marlin_q, layer.marlin_s, layer.marlin_z,
                                      reshaped_x.shape[0], n, k)
        elif x.shape[:-1].numel() >= 256:  # awq.py:166-170: large batches dequantise once and use a dense GEMM
            out = torch.matmul(reshaped_x, ops.awq_dequantize(layer.qweight, layer.scales, layer.qzeros, 0, 0, 0))
        else:
            out = ops.awq_gemm(reshaped_x, layer.qweight, layer.scales, layer.qzeros, pack_factor)
        if bias is not None:
            out.add_(bias)
        return out.reshape(out_shape)
