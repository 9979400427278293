#!/usr/bin/env python3
"""First-call cost of the torch device ops the setup still uses (fresh process): which ones are worth replacing."""
import time, torch
dev = "cuda:0"
torch.zeros(1, device=dev); torch.cuda.synchronize()
def t(label, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); d1 = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); d2 = time.perf_counter() - t0
    print("%-40s first %8.2f ms   second %7.3f ms" % (label, d1 * 1e3, d2 * 1e3), flush=True)
x = torch.arange(1 << 20, dtype=torch.int64, device=dev)
xi = torch.arange(1 << 20, dtype=torch.int32, device=dev)
xd = torch.rand(1 << 20, dtype=torch.float64, device=dev)
t("full int64", lambda: torch.full((1 << 20,), -1, dtype=torch.int64, device=dev))
t("zeros int32", lambda: torch.zeros(4, dtype=torch.int32, device=dev))
t("mask compare (x != -1)", lambda: x != -1)
t("boolean index x[mask]", lambda: x[x != 5])
t("sort int64", lambda: torch.sort(x[:4096]).values)
t("cat", lambda: torch.cat([x[:10], x[:1]]))
t("max int32 .item()", lambda: int(xi.max()))
t("sub int32", lambda: xi[1:] - xi[:-1])
t("to uint8", lambda: xi.to(torch.uint8))
t("strided slice contiguous", lambda: xi[0:1 << 20:512].contiguous())
t("index with long tensor", lambda: xi[x[:100]])
t("long()", lambda: xi[:100].long())
t("bmm f64 16x144x144", lambda: torch.bmm(torch.rand(16, 144, 144, dtype=torch.float64, device=dev), torch.rand(16, 144, 52, dtype=torch.float64, device=dev)))
t("matmul f64 2241", lambda: torch.rand(1100, 1100, dtype=torch.float64, device=dev) @ torch.rand(1100, 1100, dtype=torch.float64, device=dev))
t("isfinite.all", lambda: bool(torch.isfinite(xd).all()))
t("abs().max()", lambda: float(xd.abs().max()))
t("eye", lambda: torch.eye(100, dtype=torch.float64, device=dev))
t("slice assign 2D", lambda: xd.view(1024, 1024)[:100, :100].copy_(xd.view(1024, 1024)[100:200, 100:200]))
t("linalg.inv 64", lambda: torch.linalg.inv(torch.eye(64, dtype=torch.float64, device=dev) * 2))
