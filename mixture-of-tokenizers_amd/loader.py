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


def _load_data_shard(file: Path, dtype: torch.dtype = torch.uint16) -> Tensor:
    """train_gpt.py:629-638: 256 x int32 header [magic, version, num_tokens], then the payload."""
    header = np.fromfile(str(file), dtype=np.int32, count=HEADER_INT32)
    assert header[0] == SHARD_MAGIC, f"magic number mismatch in the data .bin file: {header[0]}"
    assert header[1] == SHARD_VERSION, f"unsupported version, expected 1 but got {header[1]}"
    num_tokens = int(header[2])
    tokens = torch.empty(num_tokens, dtype=dtype, pin_memory=torch.cuda.is_available())
    with Path(file).open("rb", buffering=0) as f:
        f.seek(HEADER_INT32 * 4)
        nbytes = f.readinto(tokens.numpy())
    assert nbytes == num_tokens * tokens.element_size(), "number of tokens read does not match header"
    return tokens


def load_data_shard(file_iter) -> Tensor:
    """train_gpt.py:641-648: skip shards whose header is bad; int32 payload for files under bytes/."""
    while True:
        try:
            file = next(file_iter)
            dtype = torch.int32 if "bytes/" in str(file) else torch.uint16
            return _load_data_shard(file, dtype=dtype).to(torch.int32)
        except AssertionError:
            pass


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
    return _load_data_shard(Path(path), dtype=torch.int32)


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

    local_seq_len = seq_len + 1
    file_iter = iter(files)
    data, pos = load_data_shard(file_iter), 0
    while True:
        if pos + batch_size * local_seq_len + 1 >= len(data):
            newdata, pos = load_data_shard(file_iter), 0
            data = torch.cat([data, newdata])
        tokens = rank_slice(data, pos, batch_size, seq_len, rank, world_size).to(device)
        pos += batch_size * local_seq_len
        yield create(tokens)
