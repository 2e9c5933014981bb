"""Row a6: which decoded frames get hashed is part of the video hash. The rule is the reference's
frame_extract_pyav (vpdqpy/vpdqpy.py:72-77 rate -> stride, :85-101 index bookkeeping incl. bad frames)."""
from fractions import Fraction

import numpy as np
import pytest

from hvd_amd import vpdqpy


@pytest.mark.parametrize("rate,stride", [
    (None, 1), (Fraction(1, 2), 1), (0.99, 1), (Fraction(1), 1), (Fraction(30000, 1001), 30), (Fraction(24000, 1001), 24),
    (25, 25), (Fraction(5, 2), 2), (Fraction(7, 2), 4), (Fraction(3, 2), 2), (59.94, 60), (Fraction(1499, 1000), 1),
])
def test_stride_is_pythons_round_of_the_average_rate(rate, stride):
    assert vpdqpy.hashed_frame_stride(rate) == stride


def test_selection_keeps_multiples_of_the_stride():
    frames = list(range(100))
    assert list(vpdqpy.select_frames(frames, Fraction(30000, 1001))) == [0, 30, 60, 90]
    assert list(vpdqpy.select_frames(frames, None)) == frames
    assert list(vpdqpy.select_frames(frames, Fraction(1, 3))) == frames
    assert list(vpdqpy.select_frames([], 25)) == []
    assert vpdqpy.selected_frame_indices(100, 30).tolist() == [0, 30, 60, 90]
    assert vpdqpy.selected_frame_indices(0, 30).tolist() == []
    assert list(vpdqpy.Vpdq.select_frames(range(7), 3)) == [0, 3, 6]


class BadFrame(Exception):
    pass


class Decoder:
    """Iterator whose __next__ raises BadFrame at given decode indices and keeps going afterwards (the way
    container.decode() behaves for the reference's loop, vpdqpy.py:85-101)."""

    def __init__(self, n, bad):
        self.n, self.bad, self.i = n, set(bad), 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.i >= self.n:
            raise StopIteration
        i = self.i
        self.i += 1
        if i in self.bad:
            raise BadFrame(i)
        return i


def test_bad_frames_advance_the_index_but_are_not_hashed():
    # stride 3: indices 0,3,6,9 are wanted; 3 is bad -> skipped, the phase of 6 and 9 is unchanged
    assert list(vpdqpy.select_frames(Decoder(11, {3}), 3, (BadFrame,))) == [0, 6, 9]
    # a bad frame at an index that is not selected changes nothing
    assert list(vpdqpy.select_frames(Decoder(11, {4, 5}), 3, (BadFrame,))) == [0, 3, 6, 9]
    # an unlisted exception is the caller's problem (the reference lets it propagate to _phash_file, dedup.py:75-79)
    with pytest.raises(BadFrame):
        list(vpdqpy.select_frames(Decoder(5, {1}), 1))


def test_compute_hash_applies_the_rule_only_on_request(monkeypatch):
    seen = []

    class FakeHasher:
        def __init__(self, *a):
            pass

        def hash_frame(self, f):
            seen.append(bytes(f)[0])

        def finish(self):
            return "done"

    monkeypatch.setattr(vpdqpy.vpdq, "VideoHasher", FakeHasher)
    fr = np.arange(10, dtype=np.uint8).reshape(10, 1, 1) * np.ones((1, 4, 4), dtype=np.uint8)
    assert vpdqpy.Vpdq.computeHash(fr) == "done" and seen == list(range(10))
    seen.clear()
    vpdqpy.Vpdq.computeHash(fr, average_rate=Fraction(3), all_decoded_frames=True)
    assert seen == [0, 3, 6, 9]
    seen.clear()
    vpdqpy.Vpdq.computeHash((bytes([i]) * 16 for i in range(10)), width=4, height=4, average_rate=4, all_decoded_frames=True)
    assert seen == [0, 4, 8]
