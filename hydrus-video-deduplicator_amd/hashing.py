"""Mirror of the reference facade ``hashing.py:14-53``."""

from __future__ import annotations

from .vpdqpy import Vpdq, VpdqHash


def compute_phash(video, num_threads: int = 0) -> VpdqHash:
    """Calculate the perceptual hash of a video (pre-decoded frames; hashing.py:14-21)."""
    return Vpdq.computeHash(video, num_threads)


def encode_phash_to_str(phash: VpdqHash) -> str:
    """hashing.py:24-31"""
    return str(phash)


def decode_phash_from_str(phash_str: str) -> VpdqHash:
    """hashing.py:34-40"""
    return VpdqHash.from_string(phash_str)


def get_phash_similarity(hash_a: VpdqHash, hash_b: VpdqHash) -> float:
    """hashing.py:43-53"""
    similarity = Vpdq.match_hash(query_features=hash_a, target_features=hash_b)
    assert similarity >= 0.0 and similarity <= 100.0
    return similarity
