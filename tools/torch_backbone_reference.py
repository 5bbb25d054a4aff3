"""A yardstick, not a dependency: what plain PyTorch-ROCm (vendor BLAS + ATen kernels, autocast) needs for the dominant part
of the step - the 12 ViT-B/16 blocks over the three stacked modalities (3 x 128 sequences of 129 tokens), forward + backward,
written the way the reference writes them (vit_pytorch.py:139-145,184-198,215-220: explicit q k^T softmax and the attention
maps handed back for the rollout) - beside this repo's WHOLE training step (patch embedding, the same 12 blocks, SFTS, HMA,
loss head, backward, SGD) on the same GPU.  Nothing on the product path uses any of this.
    python tools/torch_backbone_reference.py"""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Attention(nn.Module):
    def __init__(self, dim=768, heads=12):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(b, n, c)
        return self.proj(x), attn


class Block(nn.Module):
    def __init__(self, dim=768, heads=12, ratio=4.0):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.fc1, self.fc2, self.act = nn.Linear(dim, int(dim * ratio)), nn.Linear(int(dim * ratio), dim), nn.GELU()

    def forward(self, x):
        y, attn = self.attn(self.norm1(x))
        x = x + y
        return x + self.fc2(self.act(self.fc1(self.norm2(x)))), attn


def main():
    dev = "cuda"
    torch.manual_seed(0)
    blocks = nn.ModuleList([Block() for _ in range(12)]).to(dev)
    opt = torch.optim.SGD(blocks.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    x0 = torch.randn(3 * 128, 129, 768, device=dev)
    for dtype, name in ((torch.bfloat16, "bf16"), (torch.float16, "f16 (the reference's autocast dtype, engine/processor.py:79)")):
        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=dtype):
                x, maps = x0, []
                for blk in blocks:
                    x, a = blk(x)
                    maps.append(a.detach())
                loss = x.float().square().mean()
            loss.backward()
            opt.step()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print("plain PyTorch-ROCm, autocast %s: 12 ViT-B blocks over 3 x 128 x 129 tokens, forward + backward + SGD: %.1f ms"
              "  (= %.0f tri-modal img/s if the rest of the step were free)" % (name, ms, 128 / ms * 1e3), flush=True)
        print("YARDSTICK %s %.3f" % (name.split()[0], ms), flush=True)


if __name__ == "__main__":
    main()
