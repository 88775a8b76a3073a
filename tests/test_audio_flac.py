"""Row f3 (audio ingest): the native FLAC decoder (diarizen_amd/csrc/flac.cpp, C ABI dzn_flac_info / dzn_flac_decode) behind
diarizen_amd.audio.load_flac / load_audio.  CPU only (host code of libdzn_hip.so).

No libFLAC / flac binary exists in this image, so most streams are made by testkit/flac_encoder.py, written from the same
specification — what those tests hold is the bit stream syntax construct by construct (every subframe type, Rice method, stereo
mode, header coding), the CRC-8 / CRC-16 / MD5 integrity checks, and exact sample equality with the WAV path on the reference's
own example recording.  (r6) Independent evidence: the three example files of RFC 9639, appendix D (made by libFLAC 1.3.3,
tests/golden/flac_rfc9639_example*.flac) decode to the samples the RFC walks through, with their CRCs and STREAMINFO MD5s
verified — verbatim + stereo, fixed predictors + Rice partitions + mid-side + metadata blocks, 8-bit LPC."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _signal(n, seed=0, amp=9000):
    r = np.random.default_rng(seed)
    t = np.arange(n)
    return (amp * np.sin(t * 0.031) + 0.3 * amp * np.sin(t * 0.23 + 1.0) + 0.02 * amp * r.standard_normal(n)).astype(np.int64)


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("po,rice5", [(0, False), (3, False), (2, True)])
def test_fixed_predictors_and_rice_partitions(built_lib, order, po, rice5):
    from diarizen_amd.audio import load_flac
    from testkit.flac_encoder import FrameSpec, SubSpec, encode
    x = _signal(5000, order)
    data = encode(x, 16000, 16, frames=[FrameSpec(blocksize=1024, subs=[SubSpec("fixed", order, partition_order=po, rice5=rice5)])])
    y, sr = load_flac(data)
    assert sr == 16000 and y.shape == (1, 5000) and y.dtype == np.float32
    assert np.array_equal(y[0], (x / 32768.0).astype(np.float32))


@pytest.mark.parametrize("mode", ["independent", "left_side", "side_right", "mid_side"])
def test_stereo_decorrelation_and_lpc(built_lib, mode):
    from diarizen_amd.audio import first_channel_16k, load_flac
    from testkit.flac_encoder import FrameSpec, SubSpec, encode
    left = _signal(9000, 1)
    right = (0.6 * left).astype(np.int64) + _signal(9000, 2, amp=500)
    st = np.stack([left, right], 1)
    subs = [SubSpec("fixed", 2, partition_order=2), SubSpec("lpc", 3, coefs=[2000, -1100, 100], precision=12, shift=10, rice5=True)]
    data = encode(st, 16000, 16, frames=[FrameSpec(blocksize=4096, stereo=mode, subs=subs), FrameSpec(blocksize=576, stereo=mode, subs=subs[::-1])])
    y, sr = load_flac(data)
    assert y.shape == (2, 9000) and np.array_equal(y, (st.T / 32768.0).astype(np.float32))
    assert np.array_equal(first_channel_16k(data), (left / 32768.0).astype(np.float32))      # channel 0 kept (inference.py:128)


def test_24_bit_lpc_wasted_bits_verbatim_escape_explicit_header_fields(built_lib):
    from diarizen_amd.audio import load_flac
    from testkit.flac_encoder import FrameSpec, SubSpec, encode
    r = np.random.default_rng(3)
    x = (r.integers(-2 ** 19, 2 ** 19, 6801) * 16).astype(np.int64)          # 4 wasted bits, full-range noise: long unary runs
    frames = [FrameSpec(blocksize=1152, subs=[SubSpec("lpc", 8, coefs=list(r.integers(-300, 300, 8)), precision=10, shift=9, wasted=4,
                                                       partition_order=1)], explicit_rate=True),
              FrameSpec(blocksize=777, subs=[SubSpec("verbatim")]),
              FrameSpec(blocksize=200, subs=[SubSpec("fixed", 1, escape=True)]),
              FrameSpec(blocksize=4096, subs=[SubSpec("fixed", 2)])]
    y, sr = load_flac(encode(x, 48000, 24, frames=frames))
    assert sr == 48000 and np.array_equal(y[0], (x / 8388608.0).astype(np.float32))


def test_constant_subframes_8_bit_and_frame_numbers_beyond_one_byte(built_lib):
    from diarizen_amd.audio import load_flac
    from testkit.flac_encoder import FrameSpec, SubSpec, encode
    z = np.zeros(16 * 300, np.int64)
    z[16 * 150:] = -7
    y, sr = load_flac(encode(z, 8000, 8, frames=[FrameSpec(blocksize=16, subs=[SubSpec("constant")])]))     # 300 frames: 2-byte frame numbers
    assert sr == 8000 and np.array_equal(y[0], (z / 128.0).astype(np.float32))


def test_integrity_checks(built_lib):
    from diarizen_amd.audio import load_audio, load_flac
    from testkit.flac_encoder import FrameSpec, SubSpec, encode
    x = _signal(3000, 5)
    data = encode(x, 16000, 16, frames=[FrameSpec(blocksize=1000, subs=[SubSpec("fixed", 2)])])
    assert np.array_equal(load_audio(data)[0][0], (x / 32768.0).astype(np.float32))
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x04                                              # a flipped bit inside a frame: CRC-16 (or syntax)
    with pytest.raises(ValueError, match="FLAC decode failed"):
        load_flac(bytes(bad))
    bad = bytearray(data)
    bad[4 + 4 + 18] ^= 0xff                                                 # first byte of the STREAMINFO MD5
    with pytest.raises(ValueError, match="MD5"):
        load_flac(bytes(bad))
    assert load_flac(bytes(bad), verify_md5=False)[0].shape == (1, 3000)
    no_total = encode(x, 16000, 16, frames=[FrameSpec(blocksize=1000, subs=[SubSpec("fixed", 2)])], total_in_header=False)
    assert np.array_equal(load_flac(no_total)[0][0], (x / 32768.0).astype(np.float32))   # unknown length: decode to the data end
    with pytest.raises(ValueError):
        load_flac(data[:200])                                               # truncated


def test_other_containers_are_refused_by_name(built_lib):
    from diarizen_amd.audio import load_audio
    for head, name in ((b"OggS" + bytes(60), "Ogg"), (b"ID3\x03" + bytes(60), "MP3"), (b"\xff\xfb\x90\x00" + bytes(60), "MP3"),
                       (b"\x00\x00\x00\x20ftypM4A " + bytes(60), "MP4"), (b"FORM" + bytes(60), "AIFF"), (b"NIST_1A\n" + bytes(60), "SPHERE")):
        with pytest.raises(ValueError, match=name):
            load_audio(head)
    with pytest.raises(ValueError, match="unrecognised"):
        load_audio(bytes(64))


def test_flac_of_the_reference_example_equals_its_wav(built_lib):
    """tests/golden/EN2002a_30s.wav (the reference's example recording) re-coded as FLAC gives sample-for-sample the waveform
    the WAV path gives: the pipeline cannot tell the containers apart."""
    from diarizen_amd.audio import first_channel_16k, load_wav
    from testkit.flac_encoder import FrameSpec, SubSpec, encode
    wav = os.path.join(GOLD, "EN2002a_30s.wav")
    x, sr = load_wav(wav)
    pcm = np.round(x[0, :160000] * 32768.0).astype(np.int64)                # 10 s are enough (the encoder is pure Python)
    data = encode(pcm, sr, 16, frames=[FrameSpec(blocksize=4096, subs=[SubSpec("fixed", 2, partition_order=4)])])
    assert len(data) < 0.7 * 2 * len(pcm)                                   # it does compress
    assert np.array_equal(first_channel_16k(data), x[0, :160000])


def test_unknown_length_stream_of_digital_silence_and_a_lying_total(built_lib):
    """(r6) a stream without a STREAMINFO total (piped encoder) whose frames are CONSTANT subframes codes 4096 samples
    in ~14 bytes: the loader must grow its output instead of calling the file corrupt; a STREAMINFO total that the byte count
    cannot hold is refused before anything is allocated."""
    from diarizen_amd.audio import load_flac
    from testkit.flac_encoder import FrameSpec, SubSpec, encode
    x = np.zeros(4096 * 40, dtype=np.int64)
    spec = [FrameSpec(blocksize=4096, subs=[SubSpec("constant")])]
    data = encode(x, 16000, 16, frames=spec)
    assert len(data) * 4 < len(x)                            # under the loader's first guess of 4 samples per byte
    y, sr = load_flac(data)
    assert sr == 16000 and y.shape == (1, len(x)) and not y.any()
    unknown = encode(x, 16000, 16, frames=spec, md5=False, total_in_header=False)
    y2, _ = load_flac(unknown)
    assert y2.shape == y.shape and not y2.any()
    # STREAMINFO body starts at byte 8: min/max block (4), min/max frame (6), then 20 bits rate | 3 ch | 5 bps | 36 bits total
    lying = bytearray(data)
    off = 8 + 10
    v = int.from_bytes(lying[off:off + 8], "big")
    lying[off:off + 8] = (v | ((1 << 36) - 1)).to_bytes(8, "big")          # total = 2^36 - 1 samples
    with pytest.raises(ValueError, match="holds at most"):
        load_flac(bytes(lying))


# ---- r6: streams of an INDEPENDENT encoder: the example files of RFC 9639, appendix D (libFLAC 1.3.3) ----
RFC_EXAMPLES = {
    # D.1: one frame of one inter-channel sample, 16-bit stereo 44.1 kHz, verbatim subframes
    1: dict(sr=44100, bits=16, samples=[[25588], [10416]]),
    # D.2: two frames (16 + 3 samples), left/side + right/side decorrelation, fixed predictors, Rice partitions; SEEKTABLE,
    #      VORBIS_COMMENT and PADDING blocks before the audio
    2: dict(sr=44100, bits=16, samples=[
        [10372, 18041, 14942, 17876, 15627, 17899, 16242, 18077, 16824, 18263, 17295, -14418, -15201, -14508, -15195, -14818,
         -15486, -15349, -16054],
        [6070, 10545, 8743, 10449, 9143, 10463, 9502, 10569, 9840, 10680, 10113, -8428, -8895, -8476, -8896, -8653, -9072,
         -8958, -9410]]),
    # D.3: 24 samples, 8-bit mono 32 kHz, one LPC subframe of order 3
    3: dict(sr=32000, bits=8, samples=[[0, 79, 111, 78, 8, -61, -90, -68, -13, 42, 67, 53, 13, -27, -46, -38, -12, 14, 24, 19, 6, -4, -5,
                                        0]]),
}


@pytest.mark.parametrize("ex", [1, 2, 3])
def test_rfc9639_example_files_decode_to_the_documented_samples(built_lib, ex):
    """The byte streams are the RFC's hex dumps; the decoder checks every frame's CRC-8 / CRC-16 and (load_flac) the STREAMINFO
    MD5 of the decoded PCM — a stream made by libFLAC, not by this repo's encoder."""
    import hashlib
    from diarizen_amd.audio import load_flac
    want = RFC_EXAMPLES[ex]
    data = open(os.path.join(GOLD, f"flac_rfc9639_example{ex}.flac"), "rb").read()
    y, sr = load_flac(data)                      # verify_md5=True: raises when the MD5 of the samples differs
    full = float(1 << (want["bits"] - 1))
    got = np.rint(y * full).astype(np.int64)
    assert sr == want["sr"] and np.array_equal(got, np.asarray(want["samples"], np.int64))
    # the MD5 field itself, over the interleaved little-endian PCM (RFC 9639 8.2)
    pcm = np.asarray(want["samples"], np.int64).T.astype("<i2" if want["bits"] == 16 else "i1").tobytes()
    assert hashlib.md5(pcm).digest() == data[26:42]
    # one flipped bit inside the first frame is caught by the frame's CRC
    bad = bytearray(data)
    bad[-4] ^= 0x10
    with pytest.raises(ValueError):
        load_flac(bytes(bad))
