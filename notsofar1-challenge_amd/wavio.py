"""Dependency-free RIFF/WAVE reader and writer for the two I/O edges of the CSS path.

Counterparts of ``css/helpers.py:40 load_audio`` (soundfile.read(dtype='float32') of 7 mono files
stacked to [1, n, 7], or one file to [1, n, 1]) and ``utils/audio_utils.py:37 write_wav`` (peak
normalisation ``x * 0.99 / (max|x| + 1e-7)`` followed by soundfile's default 16-bit PCM encoding).
soundfile / libsndfile are not available in this image, hence the small codec below:
  * read: PCM 16/24/32-bit integer -> float32 scaled by 2^-(bits-1) (libsndfile's normalisation),
    IEEE float32/64 passed through;
  * write: float -> PCM16 as lrint(x * 32767) (libsndfile's float->short conversion, no clipping
    needed after the 0.99 peak normalisation).
"""
from __future__ import annotations

import os
import struct
from typing import List, Tuple

import numpy as np

NUM_MICS_MC = 7  # utils/mic_array_model.py:4 multichannel_mic_pos_xyz_cm() has 7 rows


def read_wav(path) -> Tuple[np.ndarray, int]:
    """-> (float32 samples [n] or [n, channels], sample_rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, channels, rate, _, _, bits = fmt
    if tag == 1:
        if bits == 16:
            x = np.frombuffer(payload, dtype="<i2").astype(np.float32) / np.float32(32768.0)
        elif bits == 32:
            x = (np.frombuffer(payload, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        elif bits == 24:
            b = np.frombuffer(payload, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
        elif bits == 8:
            x = (np.frombuffer(payload, dtype=np.uint8).astype(np.float32) - 128.0) / np.float32(128.0)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(payload, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    if channels > 1:
        x = x[: (x.size // channels) * channels].reshape(-1, channels)
    return x, int(rate)


def read_wav_pcm16(path):
    """-> (int16 samples [n], sample_rate) if `path` is a mono 16-bit PCM wav, else None (the device-side wav edge
    of css_inference takes the raw samples; anything else goes through read_wav)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", data[pos + 8:pos + 24])
        elif cid == b"data":
            payload = data[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None or fmt[0] != 1 or fmt[1] != 1 or fmt[5] != 16:
        return None
    return np.frombuffer(payload, dtype="<i2"), int(fmt[2])


def probe_wav_pcm16(path):
    """-> (n_samples, sample_rate, byte offset of the samples) if `path` is a mono 16-bit PCM wav whose chunks can be walked from
    its first 64 KB, else None.  With read_pcm16_payload_into: the session loop's decode, straight from the file into page-locked
    memory (read_wav_pcm16 makes three copies of the payload on its way there)."""
    with open(path, "rb") as f:
        head = f.read(65536)
        f.seek(0, os.SEEK_END)
        file_size = f.tell()
    if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt = 12, None
    while pos + 8 <= len(head):
        cid, size = head[pos:pos + 4], struct.unpack("<I", head[pos + 4:pos + 8])[0]
        if cid == b"fmt " and pos + 24 <= len(head):
            fmt = struct.unpack("<HHIIHH", head[pos + 8:pos + 24])
        elif cid == b"data":
            if fmt is None or fmt[0] != 1 or fmt[1] != 1 or fmt[5] != 16:
                return None
            size = min(size, file_size - (pos + 8))
            return size // 2, int(fmt[2]), pos + 8
        pos += 8 + size + (size & 1)
    return None


def read_pcm16_payload_into(path, offset: int, out: np.ndarray) -> None:
    """The int16 samples at `offset` of `path` into `out` (contiguous int16, little-endian host), no intermediate copy."""
    assert out.dtype == np.int16 and out.flags.c_contiguous
    with open(path, "rb", buffering=0) as f:
        f.seek(offset)
        view = memoryview(out).cast("B")
        got = 0
        while got < len(view):
            n = f.readinto(view[got:])
            if not n:
                raise ValueError(f"{path}: truncated data chunk")
            got += n


def write_pcm16_samples(path, pcm: np.ndarray, sr: int) -> None:
    """Writes already encoded int16 samples as a mono 16-bit PCM wav."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    assert pcm.ndim == 1
    dir_name = os.path.dirname(str(path))
    if dir_name:
        os.makedirs(dir_name, exist_ok=True)
    nbytes = pcm.size * 2
    hdr = b"RIFF" + struct.pack("<I", 36 + nbytes) + b"WAVE" + b"fmt " + \
        struct.pack("<IHHIIHH", 16, 1, 1, int(sr), int(sr) * 2, 2, 16) + b"data" + struct.pack("<I", nbytes)
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(memoryview(pcm).cast("B"))     # (no second copy of the samples)


def write_pcm16(path, samps: np.ndarray, sr: int) -> None:
    samps = np.asarray(samps)
    assert samps.ndim == 1
    pcm = np.rint(samps.astype(np.float64) * 32767.0)
    pcm = np.clip(pcm, -32768, 32767).astype("<i2")
    payload = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE" + b"fmt " + \
        struct.pack("<IHHIIHH", 16, 1, 1, int(sr), int(sr) * 2, 2, 16) + b"data" + struct.pack("<I", len(payload))
    with open(path, "wb") as f:
        f.write(hdr + payload)


def load_audio(wav_file_names: List, is_mc: bool) -> Tuple[np.ndarray, int]:
    """css/helpers.py:40-65: -> (mix_wav float32 [Batch=1, n_samples, n_channels], sr)."""
    if is_mc:
        assert len(wav_file_names) == NUM_MICS_MC, f'expecting {NUM_MICS_MC} microphones'
        audio, srs = zip(*[read_wav(p) for p in wav_file_names])
        mix_wav = np.stack(audio, axis=-1)[np.newaxis, ...]
        assert mix_wav.ndim == 3 and mix_wav.shape[2] in (1, 7)
        sr = srs[0]
    else:
        assert len(wav_file_names) == 1
        mix_wav, sr = read_wav(wav_file_names[0])
        assert mix_wav.ndim == 1
        mix_wav = mix_wav[np.newaxis, :, np.newaxis]
    return np.ascontiguousarray(mix_wav, dtype=np.float32), sr


def write_wav(fname, samps: np.ndarray, sr: int = 16000, max_norm: bool = True) -> None:
    """utils/audio_utils.py:37-49: peak-normalise to 0.99 and write a 16-bit mono wav."""
    assert samps.ndim == 1
    if max_norm:
        samps = samps * 0.99 / (np.max(np.abs(samps)) + 1e-7)
    dir_name = os.path.dirname(str(fname))
    if dir_name:
        os.makedirs(dir_name, exist_ok=True)
    write_pcm16(fname, samps, sr)
