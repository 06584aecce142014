"""Synthetic u8 RGB frames generated on the device (bench and examples).

Recipe of SURVEY.md 8d: smooth base (bilinear upsample of a random grid) + per-128x128-tile
Gaussian noise of sigma in {0, 2, 8, 24} + a per-frame brightness offset in {-60, -20, 20, 60},
clipped and rounded to u8.  It exercises every luminance- and texture-mask branch."""
from __future__ import annotations


def synthetic_frames(n: int, H: int, W: int, seed: int = 0, device="cuda"):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=device)
    sig = torch.tensor([0.0, 2.0, 8.0, 24.0], device=device)
    off = torch.tensor([-60.0, -20.0, 20.0, 60.0], device=device)
    ty, tx = (H + 127) // 128, (W + 127) // 128
    step = 8                                         # frames generated per pass (bounds scratch memory)
    for f0 in range(0, n, step):
        m = min(step, n - f0)
        grid = torch.rand((m, 3, H // 16 + 2, W // 16 + 2), generator=g, device=device) * 255.0
        base = torch.nn.functional.interpolate(grid, size=(H, W), mode="bilinear", align_corners=True)
        idx = (torch.arange(ty, device=device)[:, None] * 3 + torch.arange(tx, device=device)[None, :])
        fidx = torch.arange(f0, f0 + m, device=device)
        tile_sigma = sig[(idx[None] + fidx[:, None, None]) % 4]                       # [m, ty, tx]
        sigma = tile_sigma.repeat_interleave(128, 1).repeat_interleave(128, 2)[:, None, :H, :W]
        noise = torch.randn((m, 3, H, W), generator=g, device=device) * sigma
        img = base + noise + off[fidx % 4][:, None, None, None]
        out[f0:f0 + m] = img.clamp_(0, 255).round_().to(torch.uint8).permute(0, 2, 3, 1)
    return out
