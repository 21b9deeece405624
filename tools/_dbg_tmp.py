import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from neuralmagic_vllm_amd import _custom_ops as ops
from oracle import packing
import test_marlin_fuzz_gpu as F
DEV="cuda:0"
for (M,N,K,group) in [(13,7168,1408,-1),(13,7168,1408,128),(13,7168,1024,-1),(5,7168,1408,-1),(16,7168,1408,-1),(13,1024,1408,-1)]:
    dtype=torch.float16
    a,packed,s,w_ref=F._make(M,N,K,group,dtype,M+N+K)
    e=torch.empty(0,dtype=torch.int32,device=DEV); ws=torch.zeros(N//64*16,dtype=torch.int32,device=DEV)
    mq=ops.gptq_marlin_repack(packed,e,K,N,4); ms=packing.marlin_permute_scales(s,K,N,group)
    out=ops.gptq_marlin_gemm(a,mq,ms,e,e,ws,4,M,N,K,True)
    two=torch.empty(M,N//2,dtype=dtype,device=DEV); ops.silu_and_mul(two,out)
    one=ops.gptq_marlin_gemm_silu_and_mul(a,mq,ms,e,e,ws,4,M,N,K,True)
    torch.cuda.synchronize()
    d=(one.float()-two.float()).abs()
    bad=(one.view(torch.int16)!=two.view(torch.int16))
    print((M,N,K,group),"mismatch",int(bad.sum()),"of",bad.numel(),"max abs",float(d.max()),"rows",bad.any(1).nonzero().flatten().tolist()[:8],"cols",bad.any(0).nonzero().flatten().tolist()[:12])
