"""3x3 conv 32->32 at the first ResNet stage's geometry: dedicated kernel vs the generic split contraction."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import ops
dev = torch.device("cuda:0")
B, H, W = 256, 80, 798
torch.manual_seed(0)
img = torch.zeros(B, H + 2, W + 2, 32, device=dev)
img[:, 1:-1, 1:-1] = torch.randn(B, H, W, 32, device=dev)
w = torch.randn(32, 288, device=dev) * 0.05
bias = torch.randn(32, device=dev)
W3 = ops.split_weights(w)
out = torch.zeros_like(img)
for _ in range(2): ops.conv3x3_c32(img, W3, bias, relu=True, out=out)
torch.cuda.synchronize()
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(5): ops.conv3x3_c32(img, W3, bias, relu=True, out=out)
en.record(); torch.cuda.synchronize()
dt = st.elapsed_time(en) / 5 * 1e-3
print(f"conv3x3_c32: {dt*1e3:.3f} ms  {2.0*B*H*W*32*288/dt/1e12:.1f} TF/s", flush=True)
