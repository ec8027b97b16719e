"""TEST-ONLY stand-in for the predictor's HipBackend: same interface, computed with the CPU oracle + numpy, so the
direction / sharding / exchange logic can run under gloo on a host without a GPU.  Never imported by the package."""
import numpy as np
import torch

from oracle import predictor_numpy as P
from volume_segmantics_amd import dist as vdist


class OracleBackend:
    def __init__(self, model, vol_u8, classes, mode, want_probs):
        self.model, self.vol, self.classes, self.mode = model, vol_u8, classes, mode
        n = vol_u8.size
        self.nvox = n
        self.labels = np.zeros(n, np.uint8)
        self.probs = np.zeros(n, np.float16)
        self.keys = torch.zeros(vdist.padded_len(n, vdist.world()[1]), dtype=torch.int32)
        self._merged = None
        self.votes = torch.zeros((classes, n), dtype=torch.uint8)
        self.calls = []

    def _view(self, flat, m):
        item = flat.itemsize
        return np.lib.stride_tricks.as_strided(flat[m.base:], shape=(m.depth, m.h, m.w),
                                               strides=(m.ss * item, m.sh * item, m.sw * item)) if m.ss >= 0 and m.sh >= 0 and m.sw >= 0 else None

    def _addr(self, m, s0, nb):
        s, h, w = np.meshgrid(np.arange(s0, s0 + nb), np.arange(m.h), np.arange(m.w), indexing="ij")
        return m.base + s * m.ss + h * m.sh + w * m.sw

    def run_batch(self, m, direction, s0, nb, votes=1):
        self.calls.append((direction, s0, nb))
        addr = self._addr(m, s0, nb)
        slices = self.vol.ravel()[addr]
        x = torch.from_numpy(np.stack([P.preprocess_slice(s) for s in slices])).unsqueeze(1)
        with torch.no_grad():
            probs = torch.softmax(self.model(x), 1)
        lab = torch.argmax(probs, 1)
        mp = torch.gather(probs, 1, lab.unsqueeze(1)).squeeze(1)
        ct, cl = m.crop_top, m.crop_left
        lab = lab[:, ct:ct + m.h, cl:cl + m.w].numpy().astype(np.uint8)
        mp = mp[:, ct:ct + m.h, cl:cl + m.w].numpy().astype(np.float16)
        if self.mode == 0:
            self.labels[addr], self.probs[addr] = lab, mp
        elif self.mode == 1:
            k = self.keys.numpy().view(np.uint32)
            k[addr] = np.maximum(k[addr], P.pack_key(mp, lab, direction))
        else:
            v = self.votes.numpy()
            for c in range(self.classes):
                v[c][addr] += (lab == c).astype(np.uint8) * np.uint8(votes)

    @staticmethod
    def _unpack(keys_i32):
        l, p = P.unpack_key(keys_i32.numpy().view(np.uint32))
        return torch.from_numpy(l.copy()), torch.from_numpy(p.copy())

    def exchange(self, want_probs=True, all_ranks=True):
        """The product's exchange (dist.exchange_keys_sharded) with a numpy unpack in place of vs_keys_unpack."""
        if self.mode == 1:
            self._merged = vdist.exchange_keys_sharded(self.keys, self._unpack, want_probs, all_ranks)
        elif self.mode == 2:
            vdist.allreduce_sum_votes(self.votes)

    def results(self, shape, want_probs):
        if self.mode == 2:
            return self.votes.numpy().reshape((self.classes,) + tuple(shape)), None
        if self.mode == 1:
            if self._merged is None:
                self._merged = self._unpack(self.keys)
            l, p = (None if t is None else t.numpy()[:self.nvox] for t in self._merged)
        else:
            l, p = self.labels, self.probs
        return l.reshape(shape), (p.reshape(shape) if (want_probs and p is not None) else None)
