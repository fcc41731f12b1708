"""`-acc` frame interpolation plumbing: the reference's
`src/utils/frame_interpolation.py:11-68` around the FILM TorchScript model (`film_net_fp16.pt`, an opaque blob that is
not part of either repository), with the tensors kept on the device.

The reference walks the L-1 frame pairs one by one and, per pair, inserts `inter_frames` frames in a
closest-to-the-middle order; every model call moves its two fp32 CPU frames to the GPU (`.half().cuda()`) and its
prediction back (`.clamp(0, 1).cpu().float()`), i.e. 3 (L-1) inter_frames PCIe transfers and (L-1) inter_frames
batch-1 launches per clip.  Here the clip goes to the device once (fp16, what the model sees anyway), the insertion
order — which depends on `inter_frames` only — is computed once, and each insertion step runs the model ONCE on the
batch of all L-1 pairs.  Same arithmetic per frame (the model is batch-independent), same output layout.

The model itself is used as given (`torch.jit.load`); no kernel work happens here.
"""
import os

import torch


def init_frame_interpolation_model(checkpoint_name=os.path.join("./pretrained_model/film_net_fp16.pt"), device="cuda"):
    """src/utils/frame_interpolation.py:11-19"""
    if not os.path.isfile(checkpoint_name):
        raise FileNotFoundError(f"{checkpoint_name}: the FILM TorchScript checkpoint (README.md of the reference: "
                                "'film_net_fp16.pt') is not part of this repository")
    model = torch.jit.load(checkpoint_name, map_location="cpu")
    model.eval()
    return model.half().to(device=device)


def insertion_schedule(inter_frames):
    """[(left, right, position, numerator, denominator)] in the order the reference inserts the `inter_frames` new
    frames between two given ones (src/utils/frame_interpolation.py:32-63): at every step, of all (interval, missing
    slot) combinations the one whose slot lies closest to the interval's middle — fp32 arithmetic and first-minimum tie
    break as there.  left / right index the CURRENT result list, `position` is where the prediction is inserted;
    the model's time argument is numerator / denominator (see `_dt`)."""
    n = int(inter_frames)
    splits = torch.linspace(0, 1, n + 2)
    placed = [0, n + 1]            # slots already filled, ascending
    missing = list(range(1, n + 1))
    plan = []
    while missing:
        lo = splits[placed[:-1]]
        hi = splits[placed[1:]]
        off_centre = ((splits[None, missing] - lo[:, None]) / (hi[:, None] - lo[:, None]) - 0.5).abs()
        flat = int(torch.argmin(off_centre))
        left, pick = divmod(flat, len(missing))
        slot = missing.pop(pick)
        numer = splits[slot] - splits[placed[left]]
        denom = splits[placed[left + 1]] - splits[placed[left]]
        pos = sum(1 for s in placed if s < slot)
        placed.insert(pos, slot)
        plan.append((left, left + 1, pos, numer, denom))
    return plan


def _dt(like, numer, denom, batch):
    """the reference's `x0.new_full((1, 1), numer) / denom`: the numerator is rounded to the frames' dtype (fp16) before
    the division by the fp32 scalar"""
    return (like.new_full((1, 1), float(numer)) / denom.to(like.device)).expand(batch, 1).contiguous()


@torch.no_grad()
def batch_images_interpolation_tool(input_tensor, model, inter_frames=1, device=None, output_device="cpu",
                                    pair_chunk=16):
    """input_tensor (bs, C, F, H, W) in [0, 1] -> (bs, C, (F-1)(inter_frames+1)+1, H, W) fp32 on `output_device`
    (src/utils/frame_interpolation.py:22-68).  `device`: where the model lives (default: the model's own parameters /
    cuda).  `pair_chunk`: frame pairs per model call (FILM's full-resolution feature pyramids cost 0.1-1 GB per pair at
    512-768 px, so a long clip is walked in row slices; the schedule and the output layout do not depend on it; None =
    all pairs in one call).  Bit-equality with the reference's pairwise loop is established for a batch-independent
    stand-in model (tests/test_frame_interpolation.py); with the real FILM checkpoint (absent here) batched fp16
    convolutions may pick other MIOpen algorithms than batch 1 — untested."""
    bs, C, F, H, W = input_tensor.shape
    n = int(inter_frames)
    if device is None:
        try:
            device = next(model.parameters()).device
        except (StopIteration, AttributeError, TypeError):
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if F < 2 or n < 1:
        return input_tensor.to(output_device, torch.float32)
    frames = input_tensor.to(device=device, dtype=torch.float16)       # one upload; the model consumes fp16
    P = F - 1
    # all frame pairs at once: (bs*P, C, H, W) batches, pair p of sample b at row b*P + p
    first = frames[:, :, :-1].permute(0, 2, 1, 3, 4).reshape(bs * P, C, H, W)
    second = frames[:, :, 1:].permute(0, 2, 1, 3, 4).reshape(bs * P, C, H, W)
    results = [first, second]
    for left, right, pos, numer, denom in insertion_schedule(n):
        x0, x1 = results[left], results[right]
        rows = x0.shape[0]
        step = rows if not pair_chunk else max(1, int(pair_chunk))
        preds = [model(x0[s:s + step], x1[s:s + step], _dt(x0, numer, denom, min(step, rows - s)))
                 for s in range(0, rows, step)]
        pred = preds[0] if len(preds) == 1 else torch.cat(preds, dim=0)
        results.insert(pos, pred.clamp(0, 1).to(torch.float16))
    # (bs*P, n+1, C, H, W): each pair's first frame followed by its n inserted ones; the originals keep their fp32 values
    out = torch.empty((bs, C, P * (n + 1) + 1, H, W), dtype=torch.float32, device=output_device)
    src32 = input_tensor.to(output_device, torch.float32)
    for k, r in enumerate(results[:-1]):
        if k == 0:
            out[:, :, 0:P * (n + 1):n + 1] = src32[:, :, :-1]
        else:
            out[:, :, k:P * (n + 1):n + 1] = r.reshape(bs, P, C, H, W).permute(0, 2, 1, 3, 4).to(output_device, torch.float32)
    out[:, :, -1] = src32[:, :, -1]
    return out
