"""HBM-resident cache of the frozen conv trunk's per-frame tokens -- the MI355X counterpart of the reference's
``@torchcache(persistent=True)`` on its video backbones (``routeformer/models/video_backbone/__init__.py:14-32``,
``experiments/full_comparison.py:229-244``): the authors' steady state never re-runs the frozen image encoder on a
frame it has seen.

* key   = 64-bit content hash of the frame's bytes (``rf_frame_hash``), never a tensor address;
* table = open-addressing key -> slot map in device memory (``rf_cache_lookup`` / ``rf_cache_insert``);
* store = ``[capacity][65][240]`` tokens in HBM (62 KB per frame in fp32: a 288 GB card keeps millions of frames,
  i.e. the whole dataset after the first epoch), gathered / scattered with plain index copies;
* ``save`` / ``load`` make it persistent across runs (torchcache's ``persistent=True``).
"""
from __future__ import annotations

from typing import Optional

import torch

from routeformer_amd import _hip
from routeformer_amd._hip import check, ptr


def _stream():
    return torch.cuda.current_stream().cuda_stream


class TokenCache:
    def __init__(self, capacity_frames: int, device, tokens_per_frame: int = 65, channels: int = 240,
                 dtype=torch.float32, seed: int = 0):
        self.capacity = int(capacity_frames)
        cap = 1
        while cap < 2 * self.capacity:
            cap *= 2
        dev = torch.device(device)
        self.table_keys = torch.zeros(cap, dtype=torch.int64, device=dev)
        self.table_slots = torch.full((cap,), -1, dtype=torch.int32, device=dev)
        self.next_slot = torch.zeros(1, dtype=torch.int32, device=dev)
        self.store = torch.empty(self.capacity, tokens_per_frame, channels, dtype=dtype, device=dev)
        self._misses = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seed = seed
        self.fingerprint = 0  # digest of the producing trunk's weights (bind): part of every key's namespace
        self._ids_cache = {}
        self.hits = self.lookups = 0

    def bind(self, fingerprint: int):
        """Namespace the keys with the producing backbone's weight digest (HRNet16Backbone.fingerprint): tokens stored
        under other weights are never served.  A cache that already holds tokens of another fingerprint refuses."""
        fingerprint = int(fingerprint) & 0x7FFFFFFFFFFFFFFF
        if self.fingerprint not in (0, fingerprint) and int(self.next_slot.item()) > 0:
            raise ValueError("TokenCache holds tokens computed with other backbone weights; use a fresh cache (or load one "
                             "saved with these weights)")
        self.fingerprint = fingerprint

    def _key_seed(self) -> int:
        """seed ^ weight digest ^ arithmetic mode: the stored tokens depend on all three (a cache filled in bf16 mode must
        not answer an fp32-mode forward and vice versa)."""
        from routeformer_amd import kernels as K
        mode = 0x5BF16BF16BF16BF1 if K.get_precision() == "bf16" else 0
        return (int(self.seed) ^ self.fingerprint ^ mode) & 0x7FFFFFFFFFFFFFFF

    # -- keys -------------------------------------------------------------------------------------------------------
    def keys_of(self, clips) -> torch.Tensor:
        """clips: [(video (B,T,3,H,W) contiguous device tensor, frame idx (F,) or None)] -> int64 keys of the selected
        frames, clip-major then (b, f) -- the order ``encode_clips`` emits tokens in."""
        out = []
        for v, idx in clips:
            assert v.is_cuda and v.is_contiguous() and v.dim() == 5
            B, T = v.shape[:2]
            nbytes = v[0, 0].numel() * v.element_size()
            if idx is None:
                ids, n = None, B * T
            else:
                # frame ids of the selection, built once per (B, T, selection): a per-call `idx.to(device)` is a pageable
                # host-to-device copy, i.e. a host stall in the middle of every step that consults the cache
                key = (B, T, tuple(int(i) for i in idx.tolist()), str(v.device))
                ids = self._ids_cache.get(key)
                if ids is None:
                    ids = (torch.arange(B, device=v.device).view(B, 1) * T + idx.to(v.device).view(1, -1)).reshape(-1).contiguous()
                    self._ids_cache[key] = ids
                n = ids.numel()
            keys = torch.empty(n, dtype=torch.int64, device=v.device)
            check(_hip.lib().rf_frame_hash(ptr(v), ptr(ids), n, nbytes, ptr(keys), self._key_seed(), _stream()), "rf_frame_hash")
            out.append(keys)
        return torch.cat(out)

    # -- table ------------------------------------------------------------------------------------------------------
    def lookup(self, keys: torch.Tensor, count_misses: bool = True):
        """-> (slots int32 (n,), number of misses -- a host int, i.e. ONE device synchronisation -- or None)."""
        n = keys.numel()
        slots = torch.empty(n, dtype=torch.int32, device=keys.device)
        self._misses.zero_()
        check(_hip.lib().rf_cache_lookup(ptr(keys), n, ptr(self.table_keys), ptr(self.table_slots), self.table_keys.numel(),
                                         ptr(slots), ptr(self._misses), _stream()), "rf_cache_lookup")
        miss = int(self._misses.item()) if count_misses else None
        if miss is not None:
            self.lookups += n
            self.hits += n - miss
        return slots, miss

    def gather(self, slots: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Tokens of the given (all valid) slots, in order."""
        idx = slots.long()
        if out is None:
            return self.store.index_select(0, idx).float()
        if self.store.dtype == out.dtype:
            torch.index_select(self.store, 0, idx, out=out)
        else:
            out.copy_(self.store.index_select(0, idx))
        return out

    def insert(self, keys: torch.Tensor, slots: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        """Store ``tokens[i]`` under ``keys[i]`` for every i with ``slots[i] < 0``; returns the final slots (-1 where the
        cache is full)."""
        n = keys.numel()
        check(_hip.lib().rf_cache_insert(ptr(keys), n, ptr(self.table_keys), ptr(self.table_slots), self.table_keys.numel(),
                                         ptr(self.next_slot), self.capacity, ptr(slots), _stream()), "rf_cache_insert")
        final, _ = self.lookup(keys, count_misses=False)  # duplicates inside the batch resolve to the winner's slot
        ok = final >= 0
        self.store.index_copy_(0, final[ok].long(), tokens[ok].to(self.store.dtype))
        return final

    # -- the whole thing around a trunk pass ---------------------------------------------------------------------------
    def tokens_for(self, clips, encode, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Tokens of the selected frames: from the cache when every frame is known, else ``encode(clips, out)`` (one trunk
        pass over all of them) whose results are inserted."""
        keys = self.keys_of(clips)
        slots, miss = self.lookup(keys)
        if miss == 0:
            return self.gather(slots, out)
        tokens = encode(clips, out)
        self.insert(keys, slots, tokens)
        return tokens

    # -- persistence ----------------------------------------------------------------------------------------------------
    def state_dict(self):
        n = int(self.next_slot.item())
        return {"table_keys": self.table_keys.cpu(), "table_slots": self.table_slots.cpu(), "n": n,
                "store": self.store[:min(n, self.capacity)].cpu(), "seed": self.seed, "fingerprint": self.fingerprint}

    def load_state_dict(self, sd):
        assert sd["table_keys"].numel() == self.table_keys.numel(), "cache saved with another capacity"
        saved = int(sd.get("fingerprint", 0))
        if self.fingerprint and saved and saved != self.fingerprint:
            raise ValueError("this cache file was written with other backbone weights (fingerprint mismatch)")
        self.fingerprint = self.fingerprint or saved
        self.table_keys.copy_(sd["table_keys"])
        self.table_slots.copy_(sd["table_slots"])
        n = min(int(sd["n"]), self.capacity)
        self.next_slot.fill_(n)
        self.store[:n].copy_(sd["store"][:n])
        self.seed = sd["seed"]

    def save(self, path: str):
        torch.save(self.state_dict(), path)

    def load(self, path: str):
        self.load_state_dict(torch.load(path))
        return self
