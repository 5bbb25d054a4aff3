"""Input pipeline pieces around the hot path - row N3 of SURVEY.md 8(f).

Host side (Python, as in the reference) + one HIP kernel (csrc/augment.hip):
    RandomIdentitySampler      data/datasets/sampler.py:7-66      same class name / arguments / RNG consumption: the same
                               `random` + `numpy.random` seeds give the same index list (golden-pinned)
    ErasingParams              the rectangle selection of RandomErasing._erase (make_dataloader.py:108-130): same draws
                               from Python's `random`, returned as numbers instead of applied (golden-pinned)
    DeviceTrainTransform       T.RandomHorizontalFlip -> T.Pad -> T.RandomCrop -> T.ToTensor -> T.Normalize ->
                               RandomErasing(mode='pixel') of make_dataloader.py:245-253 for a whole batch in one launch
Not here: JPEG decode and T.Resize(interpolation=3) (PIL bicubic) stay on the host / in the decoder; the transform takes
decoded, resized uint8 (B,H,W,3) images.  The flip / crop draws use torch's CPU generator the way torchvision 0.14 does
(`torch.rand(1) < p`; `torch.randint(0, h - th + 1)`, then `w`), but torchvision is not installed in the build image, so
that ORDER is restated from its documentation, not pinned ("parity unpinned" for those two draws only).
"""
import copy
import ctypes
import math
import random
from collections import defaultdict

import numpy as np
import torch

from ._lib import call


class RandomIdentitySampler(torch.utils.data.sampler.Sampler):
    """Randomly sample N identities, then K instances of each: batch = N*K (data/datasets/sampler.py:7-66).
    data_source: list of (img_path, pid, camid, trackid)."""

    def __init__(self, data_source, batch_size, num_instances):
        self.data_source = data_source
        self.batch_size = batch_size
        self.num_instances = num_instances
        self.num_pids_per_batch = self.batch_size // self.num_instances
        self.index_dic = defaultdict(list)
        for index, (_, pid, _, _) in enumerate(self.data_source):
            self.index_dic[pid].append(index)
        self.pids = list(self.index_dic.keys())
        self.length = 0
        for pid in self.pids:
            num = max(len(self.index_dic[pid]), self.num_instances)
            self.length += num - num % self.num_instances

    def __iter__(self):
        per_pid = defaultdict(list)
        for pid in self.pids:
            idxs = copy.deepcopy(self.index_dic[pid])
            if len(idxs) < self.num_instances:
                idxs = np.random.choice(idxs, size=self.num_instances, replace=True)
            random.shuffle(idxs)
            chunk = []
            for idx in idxs:
                chunk.append(idx)
                if len(chunk) == self.num_instances:
                    per_pid[pid].append(chunk)
                    chunk = []
        avai = copy.deepcopy(self.pids)
        final = []
        while len(avai) >= self.num_pids_per_batch:
            for pid in random.sample(avai, self.num_pids_per_batch):
                final.extend(per_pid[pid].pop(0))
                if len(per_pid[pid]) == 0:
                    avai.remove(pid)
        return iter(final)

    def __len__(self):
        return self.length


class ErasingParams:
    """RandomErasing(probability, mode='pixel', max_count=1)._erase's rectangle choice for ONE image
    (make_dataloader.py:108-130), consuming Python's `random` exactly as the reference does.
    -> (erase, top, left, h, w)."""

    def __init__(self, probability=0.5, min_area=0.02, max_area=1 / 3, min_aspect=0.3, max_aspect=None):
        self.probability = probability
        self.min_area, self.max_area = min_area, max_area
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))

    def __call__(self, img_h, img_w):
        if random.random() > self.probability:
            return (0, 0, 0, 0, 0)
        area = img_h * img_w
        for _ in range(10):
            target_area = random.uniform(self.min_area, self.max_area) * area
            aspect_ratio = math.exp(random.uniform(*self.log_aspect_ratio))
            h = int(round(math.sqrt(target_area * aspect_ratio)))
            w = int(round(math.sqrt(target_area / aspect_ratio)))
            if w < img_w and h < img_h:
                top = random.randint(0, img_h - h)
                left = random.randint(0, img_w - w)
                return (1, top, left, h, w)
        return (0, 0, 0, 0, 0)


class DeviceTrainTransform:
    """The train transform of make_dataloader.py:245-253 after the resize, for a batch, on the device."""

    def __init__(self, size, prob=0.5, padding=10, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), re_prob=0.5):
        self.h, self.w = size
        self.prob, self.padding = prob, padding
        self.mean = (ctypes.c_float * 3)(*mean)
        self.std = (ctypes.c_float * 3)(*std)
        self.erasing = ErasingParams(re_prob)

    def draw(self, batch):
        """Per-image parameter rows in the order the reference's per-image chain consumes its generators."""
        rows = []
        for _ in range(batch):
            flip = int(torch.rand(1).item() < self.prob)
            top = int(torch.randint(0, 2 * self.padding + 1, size=(1,)).item())
            left = int(torch.randint(0, 2 * self.padding + 1, size=(1,)).item())
            rows.append((flip, top, left) + self.erasing(self.h, self.w))
        return torch.tensor(rows, dtype=torch.int32)

    def __call__(self, images_u8, params=None, noise=None, seed=0):
        """images_u8: uint8 (B,H,W,3) on the device.  params: int32 (B,8) from draw() (drawn here if None).
        noise: optional fp32 (B,3,H,W) N(0,1) fill for the erased rectangles (else generated on the device)."""
        if not images_u8.is_cuda:
            raise RuntimeError("DeviceTrainTransform: images are not on the GPU (no CPU fallback)")
        b, h, w, c = images_u8.shape
        assert (h, w, c) == (self.h, self.w, 3) and images_u8.dtype == torch.uint8
        if params is None:
            params = self.draw(b)
        params = params.to(device=images_u8.device, dtype=torch.int32).contiguous()
        out = torch.empty(b, 3, h, w, dtype=torch.float32, device=images_u8.device)
        call("editor_augment_u8", images_u8.contiguous(), params, b, h, w, int(self.padding), self.mean, self.std, noise,
             int(seed) & 0xFFFFFFFFFFFFFFFF, out)
        return out
