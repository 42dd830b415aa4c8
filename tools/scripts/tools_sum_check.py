"""local costs of the packed table exchange for a 16 GB table (what 8 GPUs would each do)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ngs_barcode_count_amd import distributed as bcdist
n, world = 4_000_000_000, 8
table = (torch.rand(n // 8, device="cuda") < 0.2).to(torch.int32).repeat(8) * 3
table[12345] = 70000
torch.cuda.synchronize()
for rep in range(2):
    t = time.time(); small, oi, ov = bcdist.pack_table(table); torch.cuda.synchronize()
    print("pack 16 GB -> 4 GB: %.1f ms, overflow entries %d" % ((time.time() - t) * 1e3, oi.numel()))
    t = time.time(); part = bcdist.sum_slices(small, world, n // world, torch.int32); torch.cuda.synchronize()
    print("sum of 8 byte slices -> int32 slice: %.1f ms" % ((time.time() - t) * 1e3))
    t = time.time(); s2, _, _ = bcdist.pack_table(part); torch.cuda.synchronize()
    print("pack slice: %.1f ms" % ((time.time() - t) * 1e3))
    t = time.time(); table.copy_(small); torch.cuda.synchronize()
    print("unpack u8 -> int32 (16 GB written): %.1f ms; peak %.0f GB" % ((time.time() - t) * 1e3, torch.cuda.max_memory_allocated() / 1e9))
