"""CPU: the streaming-Paraformer oracle is pinned against goldens minted from the real reference classes
(oracle/gen_golden_paraformer_streaming.py: PARAFORMER_ENCODER / PARAFORMER_DECODER of Export_Paraformer_Streaming.py)."""
import numpy as np
import pytest

from conftest import sub
from helpers import kaldi_audio, load_golden
from oracle.paraformer_streaming_oracle import ParaformerStreamingOracle

F32_TOL = 2e-4


def streaming_setup(g):
    cfg = getattr(sub("config"), str(g["cfg_name"]))()
    ck = sub("checkpoints").synth_paraformer_checkpoint(cfg, int(g["ckpt_seed"]))
    if not np.isnan(float(g["cif_bias"])):
        ck["predictor.cif_output.bias"] = np.asarray([float(g["cif_bias"])], dtype=np.float32)
    return cfg, ck


def streaming_cases(g):
    for i in range(int(g["n_cases"])):
        p = f"c{i}_"
        c = {k[len(p):]: g[k] for k in g if k.startswith(p) and not k[len(p):].startswith("k")}
        c["chunks"] = [{k[len(f"{p}k{j}_"):]: g[k] for k in g if k.startswith(f"{p}k{j}_")} for j in range(int(g[p + "n_chunks"]))]
        yield i, c


@pytest.mark.parametrize("fixture", ["paraformer_streaming_tiny", "paraformer_streaming_sparse", "paraformer_streaming_large"])
def test_streaming_oracle_matches_reference_goldens(fixture):
    g = load_golden(fixture)
    cfg, ck = streaming_setup(g)
    orc = ParaformerStreamingOracle(cfg, ck, chunk=int(g["chunk"]))
    assert (orc.B, orc.C, orc.en_keep, orc.de_keep, orc.fsmn_hist) == (9, 4, 36, 9, 10)
    for i, c in streaming_cases(g):
        audio = kaldi_audio(c["audio_seed"], int(c["n_chunks"]) * int(g["chunk"]))
        recs = orc.run(audio)
        assert [r["n"] for r in recs] == c["n_fired"].tolist(), i
        assert np.abs(np.asarray([r["cif_alphas"] for r in recs]) - c["cif_alphas"]).max() < F32_TOL
        small = cfg.d_model <= 128
        for r, k in zip(recs, c["chunks"]):
            enc = r["enc_out"] if small else r["enc_out"][:, ::8]
            assert np.abs(enc - k["enc_out"]).max() < F32_TOL
            if r["n"]:
                lg = r["logits"] if small else r["logits"][:, ::37]
                assert np.abs(lg - k["logits"]).max() < F32_TOL
                if small:
                    assert np.abs(r["list_frame"] - k["list_frame"]).max() < F32_TOL
        toks = np.concatenate([r["token_ids"] for r in recs])
        assert np.array_equal(toks, c["token_ids"])
    if fixture == "paraformer_streaming_sparse":
        assert (np.concatenate([c["n_fired"] for _, c in streaming_cases(g)]) == 0).any()      # decoder skipped on silent chunks
