"""Input pipeline pieces around the hot path - row N3 of SURVEY.md 8(f).

Host side (Python, as in the reference) + one HIP kernel (csrc/augment.hip):
    RandomIdentitySampler      data/datasets/sampler.py:7-66      same class name / arguments / RNG consumption: the same
                               `random` + `numpy.random` seeds give the same index list (golden-pinned)
    RandomIdentitySampler_DDP  data/datasets/sampler_ddp.py:111-196  per-rank mini-batches of one agreed global list
    ErasingParams              the rectangle selection of RandomErasing._erase (make_dataloader.py:108-130): same draws
                               from Python's `random`, returned as numbers instead of applied (golden-pinned)
    DeviceTrainTransform       T.RandomHorizontalFlip -> T.Pad -> T.RandomCrop -> T.ToTensor -> T.Normalize ->
                               RandomErasing(mode='pixel') of make_dataloader.py:245-253 for a whole batch in one launch
Not here: JPEG decode and T.Resize(interpolation=3) (PIL bicubic) stay on the host / in the decoder; the transform takes
decoded, resized uint8 (B,H,W,3) images.  The flip / crop draws use torch's CPU generator the way torchvision 0.14 does
(`torch.rand(1) < p`; `torch.randint(0, h - th + 1)`, then `w`), but torchvision is not installed in the build image, so
that ORDER is restated from its documentation, not pinned ("parity unpinned" for those two draws only; what the draws do to the
pixels is pinned to Pillow: tests/golden/f19_flip_pad_crop.npz).
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


class RandomIdentitySampler_DDP(torch.utils.data.sampler.Sampler):
    """data/datasets/sampler_ddp.py:111-196: every rank builds the SAME global list of identity batches from a seed
    agreed across ranks, then keeps its own mini-batches - block j of `batch_size // world` indices goes to rank
    j % world.  Same numpy.random consumption as the reference (identity draw, optional with-replacement fill,
    shuffle on an identity's first appearance), so the same shared seed gives the same per-rank index lists
    (golden-pinned for world 1 / 2 / 4).  rank / world_size default to torch.distributed's."""

    def __init__(self, data_source, batch_size, num_instances, rank=None, world_size=None, seed=None):
        import torch.distributed as dist
        self.data_source = data_source
        self.batch_size = batch_size
        self.world_size = dist.get_world_size() if world_size is None else int(world_size)
        self.rank = dist.get_rank() if rank is None else int(rank)
        self.num_instances = num_instances
        self.mini_batch_size = self.batch_size // self.world_size
        self.num_pids_per_batch = self.mini_batch_size // self.num_instances
        self.index_dic = defaultdict(list)
        for index, (_, pid, _, _) in enumerate(self.data_source):
            self.index_dic[pid].append(index)
        self.pids = list(self.index_dic.keys())
        self.seed = seed
        total = 0
        for pid in self.pids:
            num = max(len(self.index_dic[pid]), self.num_instances)
            total += num - num % self.num_instances
        self.length = total // self.world_size

    def _shared_seed(self):
        """sampler_ddp.py:100-109: every rank draws, rank 0's draw wins."""
        import torch.distributed as dist
        mine = int(np.random.randint(2 ** 31))
        if self.seed is not None:
            return int(self.seed)
        if self.world_size == 1 or not (dist.is_available() and dist.is_initialized()):
            return mine
        box = [mine]
        dist.broadcast_object_list(box, src=0)
        return int(box[0])

    def global_list(self):
        """The batch-ordered index list every rank agrees on (sampler_ddp.py:166-190)."""
        k = self.num_instances
        live = copy.deepcopy(self.pids)
        queue = {}
        out = []
        while len(live) >= self.num_pids_per_batch:
            for pid in np.random.choice(live, self.num_pids_per_batch, replace=False).tolist():
                q = queue.get(pid)
                if q is None:
                    q = copy.deepcopy(self.index_dic[pid])
                    if len(q) < k:
                        q = np.random.choice(q, size=k, replace=True).tolist()
                    np.random.shuffle(q)
                    queue[pid] = q
                out.extend(q[:k])
                del q[:k]
                if len(q) < k:
                    live.remove(pid)
        return out

    def __iter__(self):
        np.random.seed(self._shared_seed())
        allidx = np.asarray(self.global_list(), dtype=np.int64)
        total, mini = len(allidx), self.mini_batch_size
        blocks = (int(math.ceil(total / self.world_size)) // mini)
        pos = ((np.arange(blocks) * self.world_size + self.rank)[:, None] * mini + np.arange(mini)[None, :]).reshape(-1)
        mine = allidx[pos[pos < total]].tolist()
        self.length = len(mine)
        return iter(mine)

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


def resize_coeffs(in_size, out_size, interpolation=3):
    """Tap table of one axis of T.Resize on uint8 images - Pillow's precompute_coeffs + normalize_coeffs_8bpc
    (src/libImaging/Resample.c; torchvision 0.14.1 resizes PIL images with PIL.Image.resize): per output coordinate the
    window [xmin, xmin + count) and its 22-bit fixed-point weights.  interpolation: 3 = bicubic (a = -0.5, support 2; the
    train transform, make_dataloader.py:246), 2 = bilinear (support 1; the val transform, :256).  All in float64 in the
    order Pillow's C doubles take, vectorised over the output coordinates -> (bounds (out,2) int32, k (out,ksize) int32)."""
    fsupport = {2: 1.0, 3: 2.0}[interpolation]
    scale = float(in_size) / out_size
    fscale = max(scale, 1.0)
    support = fsupport * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)                 # C (int): truncation (values >= -0.5)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    j = np.arange(ksize, dtype=np.float64)[None, :]
    x = np.abs((j + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fscale))
    if interpolation == 3:
        a = -0.5
        w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))
    else:
        w = np.where(x < 1.0, 1.0 - x, 0.0)
    w = np.where(j < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for c in range(ksize):                                                          # Pillow sums the taps left to right
        ww = ww + w[:, c]
    wn = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = wn * float(1 << 22)
    kk = np.where(wn < 0, (-0.5 + fixed).astype(np.int64), (0.5 + fixed).astype(np.int64)).astype(np.int32)
    kk = np.where(j < xmax[:, None], kk, 0).astype(np.int32)
    return np.stack([xmin, xmax], axis=1).astype(np.int32), kk


class DeviceResize:
    """T.Resize(size, interpolation) for a batch of decoded uint8 (B,H,W,3) images on the device (editor_resize_u8):
    bit-exact with what the reference's PIL pipeline produces for the same pixels.  JPEG decode stays on the host."""

    def __init__(self, size, interpolation=3):
        self.size = (int(size[0]), int(size[1]))
        self.interpolation = interpolation
        self._tabs = {}

    def _tables(self, n_in, n_out, device):
        key = (n_in, n_out, device)
        t = self._tabs.get(key)
        if t is None:
            b, k = resize_coeffs(n_in, n_out, self.interpolation)
            t = self._tabs[key] = (torch.from_numpy(b).to(device).contiguous(), torch.from_numpy(k).to(device).contiguous(),
                                   int(k.shape[1]))
        return t

    def __call__(self, images_u8):
        if not images_u8.is_cuda:
            raise RuntimeError("DeviceResize: images are not on the GPU (no CPU fallback)")
        b, h, w, c = images_u8.shape
        assert c == 3 and images_u8.dtype == torch.uint8
        oh, ow = self.size
        dev = images_u8.device
        out = torch.empty(b, oh, ow, 3, dtype=torch.uint8, device=dev)
        xb, xk, xks = self._tables(w, ow, dev) if ow != w else (None, None, 1)
        yb, yk, yks = self._tables(h, oh, dev) if oh != h else (None, None, 1)
        tmp = torch.empty(b, h, ow, 3, dtype=torch.uint8, device=dev) if (ow != w and oh != h) else None
        call("editor_resize_u8", images_u8.contiguous(), b, h, w, oh, ow, xb, xk, xks, yb, yk, yks, tmp, out)
        return out


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


class DeviceJpegDecoder:
    """`Image.open(path).convert('RGB')` + the 256-wide crops of data/datasets/bases.py:9-41 for a BATCH of baseline JPEG
    files: the host Huffman-decodes each file into quantised DCT coefficients (a thread pool; the C entry point releases
    the GIL), ONE device call per geometry does dequantisation + IDCT + chroma upsampling + YCbCr -> RGB + the crop split
    (editor_jpeg_reconstruct).  Pixels are bit-identical to Pillow's (tests/golden/f14_decode.npz).

        dec = DeviceJpegDecoder(crop_w=256)
        crops = dec([open(p, "rb").read() for p in paths], device)     # uint8 (ncrop, B, H, 256, 3): RGB, NI, TI
        x = DeviceResize(cfg.INPUT.SIZE_TRAIN)(crops[0])               # ... the rest of the transform on the device

    Baseline, extended-sequential and progressive Huffman files (round 4) are covered; arithmetic-coded / lossless / 12-bit /
    4-component files raise (EDITOR_JPEG_UNSUPPORTED) and an incomplete progressive file is corrupt: no silent host fallback."""

    def __init__(self, crop_w=256, threads=8):
        from concurrent.futures import ThreadPoolExecutor
        from . import _lib
        self.crop_w = int(crop_w)
        self._cd = _lib.lib().cdll
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(threads)))

    def parse(self, data):
        """-> info (16 int32): W, H, ncomp, hmax, vmax, mcus_x, mcus_y, ycc_transform, blocks_per_image, ..."""
        buf = np.frombuffer(data, dtype=np.uint8)
        info = np.zeros(16, dtype=np.int32)
        rc = self._cd.editor_jpeg_parse(ctypes.c_void_p(buf.ctypes.data), len(data), ctypes.c_void_p(info.ctypes.data))
        if rc:
            raise ValueError("JPEG %s (editor_jpeg_parse rc %d)" % ("uses a coding mode the device decoder does not cover "
                             "(arithmetic / lossless / 12-bit / 4 components)" if rc == 9002 else "is corrupt or incomplete", rc))
        return info

    def _entropy(self, data, coef_ptr, blocks, qt_ptr, info):
        buf = np.frombuffer(data, dtype=np.uint8)
        rc = self._cd.editor_jpeg_entropy_decode(ctypes.c_void_p(buf.ctypes.data), len(data), ctypes.c_void_p(coef_ptr),
                                                 ctypes.c_long(blocks), ctypes.c_void_p(qt_ptr), ctypes.c_void_p(info.ctypes.data))
        if rc:
            raise ValueError("JPEG entropy decode failed (rc %d)" % rc)

    def __call__(self, files, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("DeviceJpegDecoder reconstructs on the GPU (no CPU fallback)")
        infos = [self.parse(f) for f in files]
        w, h = int(infos[0][0]), int(infos[0][1])
        cw = self.crop_w if self.crop_w > 0 else w
        ncrop = w // cw
        if ncrop < 1 or any(int(i[0]) != w or int(i[1]) != h for i in infos):
            raise ValueError("DeviceJpegDecoder: the files of a batch must share one image size >= the crop width")
        b = len(files)
        out = torch.empty(ncrop, b, h, cw, 3, dtype=torch.uint8, device=device)
        groups = {}
        for i, inf in enumerate(infos):                      # one launch per coefficient geometry (sampling factors)
            groups.setdefault(tuple(int(v) for v in inf[:9]), []).append(i)
        for key, idx in groups.items():
            blocks = key[8]
            n = len(idx)
            coef = torch.empty(n, blocks, 64, dtype=torch.int16).pin_memory()
            qt = torch.empty(n, 3, 64, dtype=torch.int16).pin_memory()           # (uint16 bit patterns)
            ginfo = [np.zeros(16, dtype=np.int32) for _ in idx]
            list(self._pool.map(lambda a: self._entropy(files[a[1]], coef[a[0]].data_ptr(), blocks, qt[a[0]].data_ptr(), ginfo[a[0]]),
                                enumerate(idx)))
            pb = ctypes.c_long(0)
            self._cd.editor_jpeg_planes_bytes(ctypes.c_void_p(ginfo[0].ctypes.data), ctypes.byref(pb))
            coef_d, qt_d = coef.to(device, non_blocking=True), qt.to(device, non_blocking=True)
            planes = torch.empty(n * pb.value, dtype=torch.uint8, device=device)
            dst = out if n == b else torch.empty(ncrop, n, h, cw, 3, dtype=torch.uint8, device=device)
            call("editor_jpeg_reconstruct", coef_d, qt_d, ctypes.c_void_p(ginfo[0].ctypes.data), n, planes, cw, dst)
            if n != b:
                out[:, torch.tensor(idx, device=device)] = dst
        return out
