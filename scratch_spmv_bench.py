import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import dazimsurftomo_amd as dz
dev = torch.device("cuda:0")
ctx = dz.Context(0)
# test4-joint-like: m=94317 rows, n=73440 cols, ~2548 nnz per ray row (clustered columns), + Tikhonov
m0, n, per = 94317 - 73440 + 52560, 73440, 2548
m0 = 20877
rng = np.random.default_rng(0)
g = torch.Generator(device=dev); g.manual_seed(0)
# build on device: each row = 3 blocks (dVs,Gc,Gs) x 17 layers x ~50 cells clustered
cells = 36*40
rows=[]; cols=[]
base = torch.randint(0, cells-60, (m0,), device=dev, generator=g)
k = torch.arange(50, device=dev)
lay = torch.arange(17, device=dev)
blk = torch.arange(3, device=dev)
c = (base[:,None,None,None] + k[None,None,None,:]) + lay[None,None,:,None]*cells + blk[None,:,None,None]*(cells*17)
c = c.reshape(m0,-1)
r = torch.arange(m0, device=dev)[:,None].expand_as(c)
irow = (r.reshape(-1)+1).to(torch.int32); icol=(c.reshape(-1)+1).to(torch.int32)
rw = -torch.rand(irow.shape[0], device=dev, generator=g)
# tikhonov rows
tr = torch.arange(n, device=dev)
irow = torch.cat([irow, (m0+1+tr).to(torch.int32)]); icol = torch.cat([icol, (tr+1).to(torch.int32)]); rw = torch.cat([rw, torch.full((n,), 6.0, device=dev)])
m = m0 + n; nnz = rw.shape[0]
print("m n nnz", m, n, nnz)
t0=time.time(); A = ctx.csr_from_coo(m, n, irow.contiguous(), icol.contiguous(), rw.contiguous()); torch.cuda.synchronize(); print("csr build %.3fs"%(time.time()-t0))
x = torch.randn(n, device=dev); y = torch.zeros(m, device=dev)
for it in range(5):
    ctx.aprod(1, A, x, y); s1 = ctx.kernel_seconds("spmv")
    ctx.aprod(2, A, x, y); s2 = ctx.kernel_seconds("spmvt")
bytes1 = nnz*8 + (m+1)*8 + n*4 + m*4*2; bytes2 = nnz*8 + (n+1)*8 + m*4 + n*4*2
print(f"A x : {s1*1e6:.1f} us  {bytes1/s1/1e9:.0f} GB/s   A^T y: {s2*1e6:.1f} us {bytes2/s2/1e9:.0f} GB/s")
b = torch.randn(m, device=dev); b[m0:] = 0
t0=time.time(); xs, info = ctx.lsmr(A, b, 0.01, 1e-5, 1e-4, 200, 500, 10); torch.cuda.synchronize(); dt=time.time()-t0
print(info, "lsmr %.3fs  per-it %.1f us; spmv avg %.1f us spmvt avg %.1f us"%(dt, dt/info['itn']*1e6, ctx.kernel_seconds('spmv')*1e6, ctx.kernel_seconds('spmvt')*1e6))
