"""Host -> HBM feed for the collated batches of the reference's data pipeline.

The reference's ``vqa_collate_fn`` (``src/data/dataset.py:204-251``) hands the training loop a dict with four tensors -- ``image``
[B,3,224,224] fp32 (19 MB at B = 32), ``input_ids`` / ``attention_mask`` [B,64] int64, ``label`` [B] -- next to python lists
(``question``, ``all_answers``, ``answer_counts``) that never leave the host; the loop then calls ``.to(device)`` on each tensor
(``training_pipeline.py:449-453``): a pageable copy, synchronous with the step.  ``DevicePrefetcher`` wraps any iterable of such batches:
the tensors of batch i+1 are copied on a side HIP stream while step i runs, so the 19 MB cross PCIe under the step instead of in
front of it.  The batch it yields has the same keys (tensors on the device, everything else untouched);
``as_model_inputs`` renames them to the model's keyword arguments (``image`` -> ``pixel_values``, ``label`` -> ``labels``).

Measured on an MI355X box (``profiles/r02/data_feed.log``): see DESIGN section 5 (PCIe-inclusive rate; the headline number has the
inputs resident in HBM, as the bench contract asks).
"""

from typing import Any, Dict, Iterable, Iterator, Optional

import torch

TENSOR_KEYS = ('image', 'input_ids', 'attention_mask', 'label')


def as_model_inputs(batch: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    """Collated batch -> keyword arguments of ``VietnameseVQAModel.forward`` (training_pipeline.py:449-461)."""
    out = {'pixel_values': batch['image'], 'input_ids': batch['input_ids'], 'attention_mask': batch['attention_mask']}
    if batch.get('label') is not None:
        out['labels'] = batch['label']
    return out


class DevicePrefetcher:
    """Iterates ``loader`` one batch ahead: the copies of batch i+1 are issued on a side HIP stream while step i runs, with an
    event per batch that the consumer's stream waits on.  Tensors keep the dtype the collate function gave them.

    ``pin=False`` (default): the copy is torch's pageable ``.to(device)`` -- the runtime stages it through its own pinned pool in
    chunks -- moved off the compute stream.  ``pin=True`` stages through two reusable pinned buffers of this object first; on the
    MI355X boxes of this build that is SLOWER (profiles/r02/data_feed.log: host writes into pinned -- fine-grained coherent -- memory
    run at ~1.4 GB/s, 14 ms for a 19-MB image batch), so it is only for loaders whose workers already produce pinned tensors
    (``DataLoader(pin_memory=True)``: then no extra host copy is made)."""

    def __init__(self, loader: Iterable[Dict[str, Any]], device: Optional[torch.device] = None, tensor_keys=TENSOR_KEYS, pin: bool = False):
        self.loader, self.keys, self.pin = loader, tuple(tensor_keys), pin
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.type != 'cuda':
            raise RuntimeError('DevicePrefetcher feeds a GPU; there is no CPU path on the product side')
        self.stream = torch.cuda.Stream(self.device)
        self._pinned = [{}, {}]
        self._slot_events = [None, None]               # per pinned slot: the event behind the last H2D copy that READ its buffers
        self._slot = 0

    def _stage(self, batch):
        slot = self._slot
        pin = self._pinned[slot]
        self._slot ^= 1
        out = dict(batch)
        if self.pin and self._slot_events[slot] is not None:
            # the non-blocking copy issued from this slot two batches ago may still be reading it if the side stream lags: the host
            # rewrite below must wait for it on the HOST (the consumer stream's wait_event orders device work only)
            self._slot_events[slot].synchronize()
        with torch.cuda.stream(self.stream):
            for k in self.keys:
                t = batch.get(k)
                if not isinstance(t, torch.Tensor):
                    continue
                if self.pin and not t.is_pinned():
                    buf = pin.get(k)
                    if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                        buf = pin[k] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                    buf.copy_(t)                                        # pageable -> pinned (host memcpy)
                    t = buf
                out[k] = t.to(self.device, non_blocking=True)            # -> HBM on the side stream
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._slot_events[slot] = ev
        return out, ev

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                # the pinned slot about to be rewritten belonged to the batch handed out one iteration ago (_stage waits for its copy)
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for k in self.keys:
                if isinstance(cur.get(k), torch.Tensor):
                    cur[k].record_stream(torch.cuda.current_stream(self.device))
            yield cur

    def __len__(self):
        return len(self.loader)
