import torch, math
dev=torch.device("cuda:0")
def timeit(fn,n=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
M=64300
for (N,K,kind) in [(3072,1024,"qkv"),(1024,1024,"proj"),(4096,1024,"fc1"),(1024,4096,"fc2")]:
    a=torch.randn(M,K,device=dev).bfloat16(); w=(torch.randn(N,K,device=dev)/math.sqrt(K)).bfloat16()
    bias=torch.randn(N,device=dev).bfloat16()
    ms=timeit(lambda: torch.nn.functional.linear(a,w,bias))
    print(f"torch linear (hipBLASLt) {kind} M={M} N={N} K={K}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TF/s")
