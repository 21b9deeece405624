from neuralmagic_vllm_amd.distributed.communication_op import (tensor_model_parallel_all_gather,  # noqa: F401
                                                               tensor_model_parallel_all_reduce)
from neuralmagic_vllm_amd.distributed.parallel_state import (destroy_model_parallel, get_tensor_model_parallel_rank,  # noqa: F401
                                                             get_tensor_model_parallel_world_size, get_tp_group,
                                                             init_distributed_environment)
