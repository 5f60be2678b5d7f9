"""The loader either side of the hot path: shard reading, the per-rank batch slice and the
input/target shift, mirroring ``distributed_data_generator`` of scaled-pre-train/train_gpt.py:629-806
(test copy data_loader.py:20-109).  The byte views come from the HIP kernels (data_creation.py
here); slicing/shifting are views plus ``.contiguous()`` copies, as in the reference.

Batch sharding: rank r of W takes ``data[pos + r*L : pos + (r+1)*L]`` with
``L = batch_size*(seq_len+1)/W`` and views it ``(-1, seq_len+1)`` -- rows are independent through
the whole front-end, so there is no collective on the data path.
"""
from __future__ import annotations

import functools
import random
from pathlib import Path

import numpy as np
import torch
from torch import Tensor

from .data_creation import make_embedding, pull_from_left, pull_from_right, tokens_to_bytes
from .modules import ByteHyperparameters

SHARD_MAGIC, SHARD_VERSION, HEADER_INT32 = 20240520, 1, 256


def _shard_token_count(path: Path) -> int:
    """Validates the 256 x int32 header [magic, version, num_tokens] of a shard (train_gpt.py:630-633) and returns num_tokens.
    A bad header is an AssertionError, which is what makes the stream skip the file (train_gpt.py:647)."""
    head = np.fromfile(str(path), dtype=np.int32, count=HEADER_INT32)
    if head.size < 3 or head[0] != SHARD_MAGIC:
        raise AssertionError(f"magic number mismatch in the data .bin file: {head[0] if head.size else 'empty file'}")
    if head[1] != SHARD_VERSION:
        raise AssertionError(f"unsupported version, expected 1 but got {head[1]}")
    return int(head[2])


def read_shard(path: Path, dtype: torch.dtype = torch.uint16) -> Tensor:
    """The payload of one shard (train_gpt.py:629-638) as a 1-D tensor of `dtype`, read straight into (pinned, when a GPU is
    present) host memory."""
    path = Path(path)
    n = _shard_token_count(path)
    payload = torch.empty(n, dtype=dtype, pin_memory=torch.cuda.is_available())
    with path.open("rb", buffering=0) as fh:
        fh.seek(HEADER_INT32 * 4)
        got = fh.readinto(payload.numpy())
    if got != n * payload.element_size():
        raise AssertionError("number of tokens read does not match header")
    return payload


class ShardStream:
    """The token stream ``distributed_data_generator`` walks (train_gpt.py:798-805): a growing int32 buffer plus a cursor.

    Shards are appended in the (already shuffled) order of `paths`; unreadable ones are skipped (a header AssertionError,
    train_gpt.py:641-648); files under ``bytes/`` hold int32 payloads, the others uint16 (:645).  When fewer than
    ``window + 2`` tokens remain behind the cursor the next shard is APPENDED and the cursor goes back to ZERO -- the
    reference rewinds into the tokens it has already served (``newdata, pos = ..., 0`` followed by ``cat([data, newdata])``,
    :800-802); the stream is reproduced as it is, bit for bit, not "fixed"."""

    def __init__(self, paths):
        self._paths = iter(paths)
        self.buffer = self._next_payload()
        self.cursor = 0

    def _next_payload(self) -> Tensor:
        for path in self._paths:
            try:
                wide = "bytes/" in str(path)
                return read_shard(path, torch.int32 if wide else torch.uint16).to(torch.int32)
            except AssertionError:
                continue
        # the reference's bare next() inside its generator surfaces the same way (PEP 479)
        raise RuntimeError("generator raised StopIteration: no further data shard")

    def remaining(self) -> int:
        return len(self.buffer) - self.cursor

    def advance(self, window: int) -> int:
        """Start offset of the next `window` tokens; moves the cursor past them."""
        if self.remaining() <= window + 1:
            self.buffer = torch.cat([self.buffer, self._next_payload()])
            self.cursor = 0
        start = self.cursor
        self.cursor = start + window
        return start


def write_data_shard(file: Path, tokens: np.ndarray, dtype=np.uint16) -> None:
    """Writer of the same format (data_creation.py:405-418; modded-nanogpt/data/fineweb.py:28-52)."""
    header = np.zeros(HEADER_INT32, dtype=np.int32)
    header[0], header[1], header[2] = SHARD_MAGIC, SHARD_VERSION, len(tokens)
    with open(file, "wb") as f:
        f.write(header.tobytes())
        f.write(np.asarray(tokens).astype(dtype).tobytes())


def save_file(path: str, data) -> None:
    """data_creation.py:405-418: a tensor (e.g. the (B, T, 1 + 4*bpt) batch of create_batch) as an int32 shard."""
    write_data_shard(Path(path), np.asarray(data.cpu() if isinstance(data, Tensor) else data).reshape(-1), dtype=np.int32)


def load_file(path: str) -> Tensor:
    """data_creation.py:421-459: the flat int32 payload of such a file (the caller views it (B, T, 1 + 4*bpt))."""
    return read_shard(Path(path), dtype=torch.int32)


def rank_slice(data: Tensor, pos: int, batch_size: int, seq_len: int, rank: int, world_size: int) -> Tensor:
    """train_gpt.py:795-797, 804: this rank's (batch_size/world_size, seq_len+1) rows."""
    assert batch_size % world_size == 0
    local_seq_len = seq_len + 1
    local_batch_size = (batch_size * local_seq_len) // world_size
    return data[pos + rank * local_batch_size:][:local_batch_size].view(-1, local_seq_len)


def make_create_data_from_toks(byte_params: ByteHyperparameters, ttb_in, ttb_out, pad_byte: int = 456, eot_byte: int = 457):
    """The eight ``_create_data_from_toks_*`` variants of train_gpt.py:686-783 as one function of the
    same four switches (bytes in, pull in, bytes out, pull out).  Returns
    ``(toks_in, bytes_padded_in, bytes_pulled_in, targets)`` with the reference's shapes/dtypes."""
    bpt = byte_params.bytes_per_token
    kw = dict(bytes_per_token=bpt, pad_byte=pad_byte, eot_byte=eot_byte)
    pull_in = functools.partial(pull_from_left if byte_params.padding_in == "left" else pull_from_right, **kw)
    pull_out = functools.partial(pull_from_left if byte_params.padding_out == "left" else pull_from_right, **kw)
    byte_in = byte_params.byte_mixin_method != "noop"
    byte_out = byte_params.byte_mixout_method != "noop"
    do_pull_in = byte_in and byte_params.pull_in
    do_pull_out = byte_out and byte_params.pull_out
    if (byte_in, byte_params.pull_in, byte_out, byte_params.pull_out) not in {
            (True, True, True, True), (True, False, True, True), (True, True, True, False), (True, True, False, False),
            (False, False, True, True), (False, False, True, False), (True, False, False, False), (False, False, False, False)}:
        # the reference's dispatch dict has exactly these keys (train_gpt.py:766-783)
        raise KeyError((byte_in, byte_params.pull_in, byte_out, byte_params.pull_out))

    def create_data_from_toks(toks: Tensor):
        bytes_padded_in = bytes_pulled_in = None
        if byte_in:
            bytes_padded_in = tokens_to_bytes(toks, ttb_in)
            if do_pull_in:
                bytes_pulled_in = pull_in(bytes_padded_in)[:, :-bpt].contiguous()
            bytes_padded_in = bytes_padded_in[:, :-bpt].contiguous()
        if byte_out:
            bytes_out = tokens_to_bytes(toks, ttb_out)
            if do_pull_out:
                bytes_out = pull_out(bytes_out)
            targets = bytes_out[:, bpt:].contiguous()
        else:
            targets = toks[:, 1:].contiguous()
        toks_in = toks[:, :-1].contiguous()
        return toks_in, bytes_padded_in, bytes_pulled_in, targets

    return create_data_from_toks


@torch.no_grad()
def distributed_data_generator(filename_patterns, seq_len: int, batch_size: int, rank: int, world_size: int,
                               byte_params: ByteHyperparameters, vocab_size: int = 50257, device="cuda", seed: int = 12345):
    """train_gpt.py:651-806.  Token->byte tables are read from ``embeddings/ttb_<bpt>_<side>_pad.json``
    relative to the working directory, as in the reference."""
    bpt = byte_params.bytes_per_token
    mix = byte_params.byte_mixin_method != "noop"
    need_left = mix and "left" in (byte_params.padding_in, byte_params.padding_out)
    need_right = mix and "right" in (byte_params.padding_in, byte_params.padding_out)
    ttb_left = make_embedding(f"ttb_{bpt}_left_pad.json", vocab_size).to(device) if need_left else None
    ttb_right = make_embedding(f"ttb_{bpt}_right_pad.json", vocab_size).to(device) if need_right else None
    ttb_in = ttb_left if byte_params.padding_in == "left" else ttb_right
    ttb_out = ttb_left if byte_params.padding_out == "left" else ttb_right
    create = make_create_data_from_toks(byte_params, ttb_in, ttb_out)

    if isinstance(filename_patterns, str):
        filename_patterns = [filename_patterns]
    files = sorted(Path.cwd().glob(filename_patterns[0]))
    for pattern in filename_patterns[1:]:
        files.extend(sorted(Path.cwd().glob(pattern)))
    random.seed(seed)  # all ranks shuffle the shards the same way
    random.shuffle(files)

    stream = ShardStream(files)
    window = batch_size * (seq_len + 1)           # tokens of one GLOBAL batch: every rank advances by all of it
    while True:
        start = stream.advance(window)
        yield create(rank_slice(stream.buffer, start, batch_size, seq_len, rank, world_size).to(device))
