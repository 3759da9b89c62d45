import ctypes, json, os, sys, torch
ROOT="/root/repo"
sys.path.insert(0, ROOT)
from egnn_pytorch_amd import _abi
torch.manual_seed(0)
SHAPE = sys.argv[1] if len(sys.argv) > 1 else "c4"          # c4: chain adjacency, K = 3, B 32 x N 2048; ns: plain k-NN, K = 32, B 64 x N 1024
B, N, K = (32, 2048, 3) if SHAPE == "c4" else (64, 1024, 32)
coors=torch.randn(B,N,3).cuda(); mask=torch.ones(B,N,dtype=torch.uint8).cuda()
i=torch.arange(N); adj=((i[:,None]-i[None,:]).abs()<=1).to(torch.uint8).cuda() if SHAPE == "c4" else None
idx=torch.empty(B,N,K,dtype=torch.int32).cuda(); rank=torch.empty(B,N,K).cuda()
tags=json.load(open(os.path.join(ROOT,"build_variants","index.json")))
for tag in tags:
    lib=ctypes.CDLL(os.path.join(ROOT,"build_variants",tag,"libegnn_hip.so"))
    f=lib.egnn_knn_select_f32
    f.argtypes=[ctypes.c_void_p]*3+[ctypes.c_int64]+[ctypes.c_int]*4+[ctypes.c_void_p]*3
    st=torch.cuda.current_stream().cuda_stream
    def run():
        rc=f(coors.data_ptr(),mask.data_ptr(),None if adj is None else adj.data_ptr(),0,B,N,K,3,idx.data_ptr(),rank.data_ptr(),st); assert rc==0
    for _ in range(3): run()
    torch.cuda.synchronize()
    best=1e9
    for r in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/10)
    import hashlib
    print(f"{SHAPE} {tag:40s} {best*1e3:8.1f} us  {hashlib.sha256(idx.cpu().numpy().tobytes() + rank.cpu().numpy().tobytes()).hexdigest()[:10]}")
