"""Determines, by one-hot experiments on the GPU, how v_smfmac_f32_16x16x{32,64}_f16 maps (lane, element) of the
compressed A operand + index bits and of the B operand onto dense k. Prints the maps."""
import ctypes, os, subprocess, sys
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "smfmac_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC",
                       os.path.join(here, "smfmac_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
dev = torch.device("cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())


def run(kind, a, b, idx, abid=0):
    c = torch.zeros(64, 4, dtype=torch.float32, device=dev)
    fn = L.probe32 if kind == 32 else L.probe64
    rc = fn(P(a), P(b), P(c), P(idx), abid)
    assert rc == 0
    return c.cpu()


for kind, na, nb in ((32, 4, 8), (64, 8, 16)):
    print(f"==== smfmac 16x16x{kind} f16: A {na} halfs/lane, B {nb} halfs/lane")
    # B slot id: value = 1 + slot (slot = g*nb + e), same for every column lane
    b = torch.zeros(64, nb, dtype=torch.float16)
    for g in range(4):
        for e in range(nb):
            b[g * 16:(g + 1) * 16, e] = 1 + g * nb + e
    b = b.to(dev)
    # for each A slot (ga, ja) and index value p in 0..3: which B slot does it multiply?
    for ga in range(4):
        for ja in range(na):
            row = []
            for p in range(4):
                a = torch.zeros(64, na, dtype=torch.float16)
                a[ga * 16 + 3, ja] = 1.0  # row 3
                idx = torch.zeros(64, dtype=torch.int32)
                idx[:] = p << (2 * ja)
                c = run(kind, a.to(dev), b, idx.to(dev))
                nz = c.nonzero()
                vals = sorted(set(int(v) for v in c[c != 0].tolist()))
                where = sorted(set((int(i) // 16, int(j)) for i, j in nz.tolist()))
                slot = [(v - 1) // nb for v in vals], [(v - 1) % nb for v in vals]
                row.append((p, slot, where[:2]))
            print(f"A lane-group {ga} elem {ja}:", row)
    # which D (lane, reg) holds (row r, col c)?  A row r all ones w/ idx 0 for elem0 only; B one-hot column c
    a = torch.zeros(64, na, dtype=torch.float16)
    a[5, 0] = 1.0  # row 5, lane group 0
    idx = torch.zeros(64, dtype=torch.int32)
    b2 = torch.zeros(64, nb, dtype=torch.float16)
    b2[9, 0] = 1.0  # col 9 lane group 0, k slot 0
    c = run(kind, a.to(dev), b2.to(dev), idx.to(dev))
    print("D nonzero (lane, reg) for row 5, col 9:", c.nonzero().tolist(), c[c != 0].tolist())
    # ABID: put index in byte 1 and select abid=1
    a = torch.zeros(64, na, dtype=torch.float16)
    a[3, 0] = 1.0
    for abid in range(4 if kind == 32 else 2):
        idx = torch.zeros(64, dtype=torch.int32)
        shift = (8 if kind == 32 else 16) * abid
        idx[:] = 2 << shift
        c = run(kind, a.to(dev), b, idx.to(dev), abid)
        print(f"abid {abid} (idx=2 in set {abid}) -> B slot values", sorted(set(int(v) for v in c[c != 0].tolist())))
