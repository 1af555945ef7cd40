"""The step before the hot path: length-bucketed distributed batch sampler and the padding collate of
reference data_utils.py:427-594, host-side Python like the reference's.

  DistributedBucketSampler   data_utils.py:497-594 — buckets by length boundaries, pads every bucket to a multiple of
                             num_replicas * batch_size by repetition, deals `ids[rank::num_replicas]`, shuffles buckets
                             and batches with a generator seeded by the epoch: every rank sees the same NUMBER of batches
                             of similar lengths (the step's all-reduce needs that; with the ragged rows layout the
                             similar lengths no longer save padding work, they balance the ranks).
  TextMelCollate             data_utils.py:427-494 (TextMelMyOwnCollate) — sorts by text length, right-zero-pads;
                             optionally into pinned memory, and also returns nothing the reference does not: the
                             lengths tensors it returns are what `Trainer.step(lengths_host=...)` wants as lists.
"""
import torch


class DistributedBucketSampler:
    def __init__(self, lengths, batch_size, boundaries, num_replicas=1, rank=0, shuffle=True):
        """lengths: per-sample length (the reference reads `dataset.lengths`); boundaries [b0, b1, ...]: bucket i holds
        b_i < length <= b_{i+1}; samples outside (b0, b_last] are dropped (data_utils.py:499-504)."""
        if not 0 <= rank < num_replicas:
            raise ValueError(f"rank {rank} outside [0, {num_replicas})")
        self.lengths, self.batch_size, self.boundaries = list(lengths), batch_size, list(boundaries)
        self.num_replicas, self.rank, self.shuffle, self.epoch = num_replicas, rank, shuffle, 0
        self.buckets, self.num_samples_per_bucket = self._create_buckets()
        self.total_size = sum(self.num_samples_per_bucket)
        self.num_samples = self.total_size // self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _bisect(self, x, lo=0, hi=None):
        if hi is None:
            hi = len(self.boundaries) - 1
        while hi > lo:
            mid = (hi + lo) // 2
            if self.boundaries[mid] < x <= self.boundaries[mid + 1]:
                return mid
            if x <= self.boundaries[mid]:
                hi = mid
            else:
                lo = mid + 1
        return -1

    def _create_buckets(self):
        buckets = [[] for _ in range(len(self.boundaries) - 1)]
        for i, length in enumerate(self.lengths):
            b = self._bisect(length)
            if b != -1:
                buckets[b].append(i)
        for i in range(len(buckets) - 1, 0, -1):                 # empty buckets (but never bucket 0) are merged away
            if not buckets[i]:
                buckets.pop(i)
                self.boundaries.pop(i + 1)
        total = self.num_replicas * self.batch_size
        return buckets, [len(b) + (total - len(b) % total) % total for b in buckets]

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        if self.shuffle:
            indices = [torch.randperm(len(b), generator=g).tolist() for b in self.buckets]
        else:
            indices = [list(range(len(b))) for b in self.buckets]
        batches = []
        for bucket, ids, want in zip(self.buckets, indices, self.num_samples_per_bucket):
            rem = want - len(bucket)
            ids = ids + ids * (rem // len(bucket)) + ids[:rem % len(bucket)]      # repeat to a multiple of replicas * batch
            ids = ids[self.rank::self.num_replicas]
            for j in range(len(ids) // self.batch_size):
                batches.append([bucket[k] for k in ids[j * self.batch_size:(j + 1) * self.batch_size]])
        if self.shuffle:
            batches = [batches[i] for i in torch.randperm(len(batches), generator=g).tolist()]
        self.batches = batches
        assert len(batches) * self.batch_size == self.num_samples
        return iter(batches)

    def __len__(self):
        return self.num_samples // self.batch_size


class TextMelCollate:
    """Items: (text ids [t], mel [n_mel, T]) or the fork's 8-field items (text, mel, spk_embed [512], emo, emo_cartesian [3],
    f0 [1, T], energy [1, T], lid).  Returns (text_padded, input_lengths, mel_padded, output_lengths) plus, for 8-field
    items, (spk_embeds, emos, emo_cartesians, f0_padded, energy_padded, lid) — the order of data_utils.py:494."""

    def __init__(self, n_frames_per_step=1, pin_memory=False):
        self.n_frames_per_step, self.pin = n_frames_per_step, pin_memory

    def _new(self, *shape, dtype):
        t = torch.zeros(*shape, dtype=dtype)
        return t.pin_memory() if self.pin else t

    def __call__(self, batch):
        input_lengths, order = torch.sort(torch.LongTensor([len(x[0]) for x in batch]), dim=0, descending=True)
        n = len(batch)
        text_padded = self._new(n, int(input_lengths[0]), dtype=torch.long)
        for i, k in enumerate(order.tolist()):
            text_padded[i, :batch[k][0].size(0)] = batch[k][0]
        num_mels = batch[0][1].size(0)
        max_t = max(x[1].size(1) for x in batch)
        if max_t % self.n_frames_per_step != 0:
            max_t += self.n_frames_per_step - max_t % self.n_frames_per_step
        mel_padded = self._new(n, num_mels, max_t, dtype=torch.float32)
        output_lengths = torch.zeros(n, dtype=torch.long)
        full = len(batch[0]) >= 8
        if full:
            spk = self._new(n, batch[0][2].numel(), dtype=torch.float32)
            emos, lid = torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)
            cart = self._new(n, 3, dtype=torch.float32)
            f0, energy = self._new(n, 1, max_t, dtype=torch.float32), self._new(n, 1, max_t, dtype=torch.float32)
        for i, k in enumerate(order.tolist()):
            item = batch[k]
            mel = item[1]
            mel_padded[i, :, :mel.size(1)] = mel
            output_lengths[i] = mel.size(1)
            if full:
                spk[i] = item[2]; emos[i] = item[3]; cart[i] = item[4]
                f0[i, :, :item[5].size(1)] = item[5]
                energy[i, :, :item[6].size(1)] = item[6]
                lid[i] = item[7]
        if full:
            return text_padded, input_lengths, mel_padded, output_lengths, spk, emos, cart, f0, energy, lid
        return text_padded, input_lengths, mel_padded, output_lengths
