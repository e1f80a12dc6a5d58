"""Measurement reference only (never on the product path): what the vendor library (hipBLASLt / rocBLAS through torch.matmul)
achieves on the FC shapes of the three models on this box, with full-range random operands, so that the hand-written GEMMs
can be read against it as well as against the spec peak."""
import json, sys, torch
assert torch.cuda.is_available()
dev = torch.device("cuda:0")
shapes = {  # (items M, outputs N, K)
    "C_FC1_b4096": (4096, 2048, 3968), "C_FC2_b4096": (4096, 512, 2048), "C_FC3_b4096": (4096, 256, 512),
    "B_FC1_16x1024": (16384, 1024, 880), "B_FC2_16x1024": (16384, 512, 1024), "B_FC3_16x1024": (16384, 256, 512),
    "A_FC1_64x256": (16384, 1024, 352),
    "square_8192": (8192, 8192, 8192),
}
out = {}
for name, (M, N, K) in shapes.items():
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        if tag == "f32" and name == "square_8192":
            continue
        torch.backends.cuda.matmul.allow_tf32 = False
        x = (torch.rand(M, K, device=dev) * 2 - 1).to(dt)
        w = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).to(dt)
        for _ in range(20):
            y = x @ w.t()
        torch.cuda.synchronize()
        reps = 200 if M * N * K < 4e11 else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = x @ w.t()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out[f"{name}_{tag}"] = {"us": round(us, 2), "tflops": round(2.0 * M * N * K / us / 1e6, 1)}
        print(name, tag, out[f"{name}_{tag}"], flush=True)
# fp8 (OCP e4m3) through torch._scaled_mm (hipBLASLt): per-tensor scales, bf16 output
try:
    for name in ("C_FC1_b4096", "C_FC2_b4096", "B_FC1_16x1024", "square_8192"):
        M, N, K = shapes[name]
        Kp = (K + 15) // 16 * 16
        x = ((torch.rand(M, Kp, device=dev) * 2 - 1)).to(torch.float8_e4m3fn)
        w = ((torch.rand(N, Kp, device=dev) * 2 - 1)).to(torch.float8_e4m3fn)
        one = torch.tensor(1.0, device=dev)
        f = lambda: torch._scaled_mm(x, w.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
        for _ in range(20):
            y = f()
        torch.cuda.synchronize()
        reps = 200 if M * N * K < 4e11 else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out[f"{name}_fp8"] = {"us": round(us, 2), "tflops": round(2.0 * M * N * K / us / 1e6, 1)}
        print(name, "fp8", out[f"{name}_fp8"], flush=True)
except Exception as ex:  # the fp8 path of this torch build may be missing: the bf16 / f32 rows stand on their own
    out["fp8_error"] = repr(ex)[:300]
    print("fp8 reference unavailable:", out["fp8_error"], flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
