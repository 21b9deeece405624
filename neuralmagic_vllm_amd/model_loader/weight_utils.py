"""Weight-file utilities: the parts of vllm/model_executor/model_loader/weight_utils.py the quantized hot path needs.

* `safetensors_weights_iterator` (weight_utils.py:368-376): (name, tensor) pairs of local *.safetensors files.
* `get_quant_config` (weight_utils.py:133-199): the checkpoint's quantization config from `config.json`
  (`quantization_config`, or `compression_config` for compressed-tensors) or from the method's own json file
  (`quantize_config.json` for GPTQ / AWQ / Marlin).
* `kv_cache_scales_loader` (weight_utils.py:391-428) with the checks of quantization/schema.py:18-85: per-TP-rank,
  per-layer fp8 KV-cache scaling factors from a JSON file (`--quantization-param-path`).
Nothing here downloads anything: model directories are local.
"""
import glob
import json
import logging
import os
from typing import Any, Dict, Generator, Iterable, List, Optional, Tuple

import torch

from neuralmagic_vllm_amd.layers.quantization import QuantizationConfig, get_quantization_config

logger = logging.getLogger(__name__)


def safetensors_weights_iterator(files: List[str]) -> Generator[Tuple[str, torch.Tensor], None, None]:
    from safetensors import safe_open
    for st_file in files:
        with safe_open(st_file, framework="pt") as f:
            for name in f.keys():  # noqa: SIM118
                yield name, f.get_tensor(name)


def default_weight_loader(param: torch.nn.Parameter, loaded_weight: torch.Tensor) -> None:
    assert param.size() == loaded_weight.size(), (param.size(), loaded_weight.size())
    param.data.copy_(loaded_weight)


def get_quant_config(model_dir: str, quantization: str) -> QuantizationConfig:
    quant_cls = get_quantization_config(quantization)
    hf_quant_config = None
    cfg_path = os.path.join(model_dir, "config.json")
    if os.path.isfile(cfg_path):
        with open(cfg_path) as f:
            hf_config = json.load(f)
        hf_quant_config = hf_config.get("quantization_config")
        if hf_quant_config is None:
            hf_quant_config = hf_config.get("compression_config")  # compressed-tensors
    if hf_quant_config is not None:
        return quant_cls.from_config(hf_quant_config)
    names = quant_cls.get_config_filenames()
    if not names:
        return quant_cls()
    found = [f for f in glob.glob(os.path.join(model_dir, "*.json")) if any(f.endswith(x) for x in names)]
    if len(found) == 0:
        raise ValueError(f"Cannot find the config file for {quantization}")
    if len(found) > 1:
        raise ValueError(f"Found multiple config files for {quantization}: {found}")
    with open(found[0]) as f:
        return quant_cls.from_config(json.load(f))


def _validate_kv_scales(doc: Dict[str, Any], tp_rank: int, tp_size: int, num_hidden_layers: int,
                        model_type: Optional[str]) -> Dict[int, float]:
    """The conditions of QuantParamSchema / KVCacheQuantSchema (quantization/schema.py); raises on the first violation."""
    if model_type is not None and doc.get("model_type") != model_type:
        raise ValueError(f"Model type is {model_type} but loaded scaling factors belonging to different model type "
                         f"{doc.get('model_type')}!")
    kv = doc["kv_cache"]
    if kv["dtype"] != "float8_e4m3fn":
        raise ValueError(f"Loaded scaling factors intended for KV cache dtype = {kv['dtype']} rather than float8_e4m3fn!")
    factors = {int(r): {int(layer): float(v) for layer, v in m.items()} for r, m in kv["scaling_factor"].items()}
    if len(factors) != tp_size:
        raise ValueError(f"Loaded dictionary has TP size {len(factors)} but LLM engine is currently running with TP size "
                         f"{tp_size}.")
    for r, layer_map in factors.items():
        if len(layer_map) != num_hidden_layers:
            raise ValueError(f"KV cache scales map for TP rank {r} is malformed. Expected {num_hidden_layers} layers, got "
                             f"{len(layer_map)}.")
    for r in range(tp_size):
        if r not in factors:
            raise ValueError(f"KV cache scales map for TP rank {r} not found.")
    mine = factors[tp_rank]
    for i in range(num_hidden_layers):
        if i not in mine:
            raise ValueError(f"Could not find KV cache scales for layer {i} in TP rank {tp_rank}.")
    return mine


def kv_cache_scales_loader(filename: str, tp_rank: int, tp_size: int, num_hidden_layers: int,
                           model_type: Optional[str]) -> Iterable[Tuple[int, float]]:
    """(layer index, scaling factor) pairs of this TP rank. Any error (missing file, bad JSON, failed check) is logged
    and yields nothing, which leaves every layer at the default scale 1.0 - the reference's behaviour."""
    try:
        with open(filename) as f:
            doc = json.load(f)
        return list(_validate_kv_scales(doc, tp_rank, tp_size, num_hidden_layers, model_type).items())
    except FileNotFoundError:
        logger.error("File or directory '%s' not found.", filename)
    except json.JSONDecodeError:
        logger.error("Error decoding JSON in file '%s'.", filename)
    except Exception as e:  # noqa: BLE001
        logger.error("An error occurred while reading '%s': %s", filename, e)
    logger.warning("Defaulting to KV cache scaling factors = 1.0 for all layers in TP rank %d as an error occurred during "
                   "loading.", tp_rank)
    return []
