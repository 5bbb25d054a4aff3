"""TEST INFRASTRUCTURE - CPU restatement of the train-time input transform after the resize:
T.RandomHorizontalFlip -> T.Pad(p) -> T.RandomCrop -> T.ToTensor -> T.Normalize -> RandomErasing(mode='pixel')
(data/datasets/make_dataloader.py:245-253, 55-146) with the random draws supplied as parameters.  torchvision is not
installed in the build image: flip / pad / crop / ToTensor / Normalize are restated from their documented semantics
(torchvision==0.14.1, requirements.txt:158) and pinned, for given draws, to Pillow + torch - the implementations torchvision hands a
PIL image to (tests/golden/f19_flip_pad_crop.npz, made by capture_golden.py f19; bit-equal).  "Parity unpinned" only for the ORDER of
the flip / crop draws (editor_amd/data.py); the erase rectangle and the sampler are pinned to the reference through
tests/golden/f9_input.npz.  Imported by tests/ only."""
import torch
import torch.nn.functional as F


def train_transform(images_u8, params, padding, mean, std, noise):
    """images_u8 (B,H,W,3) uint8; params (B,8) = flip, top, left, erase, e_top, e_left, e_h, e_w -> (B,3,H,W) fp32."""
    b, h, w, _ = images_u8.shape
    out = torch.empty(b, 3, h, w)
    mean_t = torch.tensor(mean).view(3, 1, 1)
    std_t = torch.tensor(std).view(3, 1, 1)
    for i in range(b):
        flip, top, left, erase, et, el, eh, ew = [int(v) for v in params[i]]
        img = images_u8[i]
        if flip:
            img = img.flip(1)                                    # PIL FLIP_LEFT_RIGHT
        img = F.pad(img.permute(2, 0, 1), (padding,) * 4)        # T.Pad: constant fill 0 on the uint8 image
        img = img[:, top:top + h, left:left + w]                 # T.RandomCrop
        x = img.float().div(255)                                 # T.ToTensor
        x = (x - mean_t) / std_t                                 # T.Normalize
        if erase:
            x[:, et:et + eh, el:el + ew] = noise[i, :, et:et + eh, el:el + ew]
        out[i] = x
    return out
