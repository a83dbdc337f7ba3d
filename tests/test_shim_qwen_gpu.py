"""GPU: Qwen3-ASR's graphs through the onnxruntime-API shim, driven by the call sequence of the reference host
(Qwen_ASR/Inference_Qwen_ASR_ONNX.py:424-745): Embed runs for the system prompt / language tail, one merged prefill launch with empty
KV tensors, then per token an Embed run into a persistent `hidden_states` buffer and one merged decode launch on alternating bindings,
state passed from outputs to inputs. Checked against the goldens minted from the reference's classes (f32 mode)."""
import json

import numpy as np
import pytest

from conftest import sub
from helpers import golden_cases, load_golden
from test_oracle_qwen_asr import qwen_setup, unit_audio

pytestmark = pytest.mark.gpu

F32, TOL = 1, 1e-3
SPECIAL = {"stop": [1, 521], "asr_text": [540], "audio_start": 524, "audio_end": 520, "audio_pad": 525, "im_start": 510, "im_end": 521, "system": 511,
           "user": 523, "assistant": 522, "newline": 512, "language_prefix": [530, 531]}
LANGS = {"en": {"name": "English", "aliases": ["english"], "prompt_token_ids": [77, 540]}, "zh": {"name": "Chinese", "aliases": [], "prompt_token_ids": [78, 540]}}


class HostLoop:
    def __init__(self, folder, strategy, repeat_penalty=1.0, penalty_range=10, sampling=(0.8, 10, 0.95, 1.0)):
        self.ort, self.io, wq = sub("ort_shim"), sub("ort_io"), sub("ort_shim_qwen")
        self.strategy, self.repeat_penalty, self.penalty_range, self.sampling = strategy, repeat_penalty, penalty_range, sampling
        self.embed = self.ort.InferenceSession(f"{folder}/{wq.EMBED_FILE}.onnx")
        self.prefill_s = self.ort.InferenceSession(f"{folder}/{wq.GRAPH_FILES['prefill_' + strategy]}.onnx")
        self.decode_s = self.ort.InferenceSession(f"{folder}/{wq.GRAPH_FILES['decode_' + strategy]}.onnx")
        self.meta = self.ort.InferenceSession(f"{folder}/{wq.METADATA_FILE}.onnx").get_modelmeta().custom_metadata_map
        names = [v.name for v in self.prefill_s.get_inputs()]
        self.kv_n = next(i for i, n in enumerate(names + ["x"]) if not n.startswith("past_"))
        self.pplan, self.dplan = self._plan(self.prefill_s, False), self._plan(self.decode_s, True)
        self.device = self.ort.OrtDevice(self.ort.OrtDevice.cuda(), self.ort.OrtDevice.default_memory(), 0)

    def _plan(self, session, is_decode):
        ins, outs = [v.name for v in session.get_inputs()], [v.name for v in session.get_outputs()]
        tail = outs[self.kv_n:]
        max_out, save_out, kv_out = (tail[0], None, tail[1]) if self.strategy == "greedy" else tail
        save_in = {"greedy": [], "sampling": ["sampling_previous_ids"]}.get(self.strategy,
                                                                             ["penalty_greedy_save_id_in"] if not is_decode else ["penalty_save_id_in", "penalty_greedy_save_id_in"])
        return dict(inputs=ins, outputs=outs, state_inputs=ins[:self.kv_n], max_out=max_out, save_out=save_out, kv_seq_out=kv_out, save_inputs=save_in,
                    meta=self.io.metadata_by_name(session.get_inputs()))

    def _ort(self, array, device="cuda"):
        return self.ort.OrtValue.ortvalue_from_numpy(np.ascontiguousarray(array), device, 0)

    def _persistent_embed(self, ids, consumer_meta):
        in_meta, out_meta = self.embed.get_inputs()[0], self.embed.get_outputs()[0]
        if not ids:
            return self._ort(self.io.filled_for(consumer_meta, axes={0: 1, 1: 0}))
        x = self.io.array_for(in_meta, [ids], axes={0: 1, 1: len(ids)})
        out = self.embed.run([out_meta.name], {in_meta.name: x})[0]
        out = self.io.array_for(out_meta, out, axes={0: 1, 1: len(ids)})
        return self._ort(self.io.array_for(consumer_meta, out, axes={0: 1, 1: len(ids)}))

    def _sampling_scalars(self, meta):
        vals = dict(zip(("sampling_temperature", "sampling_top_k", "sampling_top_p", "sampling_repetition_penalty"), self.sampling))
        return [(n, self._ort(self.io.scalar_for(meta[n], v))) for n, v in vals.items() if n in meta]

    def transcribe(self, audio, query_ids, tail_ids, max_seq_len, stop):
        pm, dm = self.pplan["meta"], self.dplan["meta"]
        prompt, tail = self._persistent_embed(query_ids, pm["query_embed"]), self._persistent_embed(tail_ids, pm["language_tail_embed"])
        out_meta = self.embed.get_outputs()[0]
        hidden = self._ort(self.io.filled_for(out_meta, axes={0: 1, 1: 1}))
        embed_binding = self.embed.io_binding()
        embed_binding.bind_ortvalue_output(out_meta.name, hidden)
        bindings = [self.decode_s.io_binding(), self.decode_s.io_binding()]
        keep = []
        for b in bindings:
            b.bind_ortvalue_input("hidden_states", hidden)
            if "penalty_penalty_value" in dm:
                for n, v in (("penalty_penalty_value", self.repeat_penalty), ("penalty_penalty_range", self.penalty_range)):
                    keep.append(self._ort(self.io.scalar_for(dm[n], v)))
                    b.bind_ortvalue_input(n, keep[-1])
            for n, v in self._sampling_scalars(dm):
                keep.append(v)
                b.bind_ortvalue_input(n, v)
        pb = self.prefill_s.io_binding()
        for name in self.pplan["state_inputs"]:
            axis = 4 if name.startswith("past_key_") else 3
            pb.bind_ortvalue_input(name, self._ort(self.io.array_for(pm[name], self.io.filled_for(pm[name], axes={0: 1, axis: 0}), axes={0: 1, axis: 0})))
        a = np.ascontiguousarray(audio, dtype=np.float32).reshape(1, 1, -1)
        pb.bind_ortvalue_input("audio", self._ort(self.io.array_for(pm["audio"], a, axes={0: 1, 1: 1, 2: a.shape[2]})))
        pb.bind_ortvalue_input("query_embed", prompt)
        pb.bind_ortvalue_input("language_tail_embed", tail)
        pb.bind_ortvalue_input("prefill_history_len", self._ort(self.io.scalar_for(pm["prefill_history_len"], 0)))
        for name in self.pplan["save_inputs"]:
            pb.bind_ortvalue_input(name, self._ort(self.io.filled_for(pm[name], axes={0: 1, 1: 0})))
        for n, v in self._sampling_scalars(pm):
            pb.bind_ortvalue_input(n, v)
        for name in self.pplan["outputs"]:
            pb._iobinding.bind_output(name, self.device)
        self.prefill_s.run_with_iobinding(pb)
        outs = pb.get_outputs()
        pos = {n: i for i, n in enumerate(self.pplan["outputs"])}
        state, kv_seq = outs[:self.kv_n], outs[pos[self.pplan["kv_seq_out"]]]
        ids_len = int(kv_seq.numpy().flat[0])
        limit = max(max_seq_len - 10 - ids_len, 0)
        next_token = outs[pos[self.pplan["max_out"]]]
        selected = int(next_token.numpy().flat[0])
        save_id = outs[pos[self.pplan["save_out"]]] if self.pplan["save_out"] else None
        tokens, count, final_save = [], 0, save_id
        if selected not in stop:
            count = 1
            if self.strategy == "greedy":
                tokens.append(selected)
        dpos = {n: i for i, n in enumerate(self.dplan["outputs"])}
        steps = 0
        while count < limit and selected not in stop:
            embed_binding.bind_ortvalue_input(self.embed.get_inputs()[0].name, next_token)
            self.embed.run_with_iobinding(embed_binding)
            b = bindings[steps & 1]
            for name, value in zip(self.dplan["state_inputs"], state):
                b.bind_ortvalue_input(name, value)
            b.bind_ortvalue_input("decode_kv_seq_len", kv_seq)
            for name in self.dplan["save_inputs"]:
                b.bind_ortvalue_input(name, save_id)
            b.clear_binding_outputs()
            for name in self.dplan["outputs"]:
                b._iobinding.bind_output(name, self.device)
            self.decode_s.run_with_iobinding(b)
            o = b.get_outputs()
            state, kv_seq, next_token = o[:self.kv_n], o[dpos[self.dplan["kv_seq_out"]]], o[dpos[self.dplan["max_out"]]]
            selected = int(next_token.numpy().flat[0])
            if self.dplan["save_out"]:
                save_id = final_save = o[dpos[self.dplan["save_out"]]]
            if selected not in stop:
                count += 1
                if self.strategy == "greedy":
                    tokens.append(selected)
            steps += 1
        if self.strategy != "greedy":
            tokens = []
            for t in final_save.numpy()[0]:
                if int(t) in stop:
                    break
                tokens.append(int(t))
        return tokens, ids_len, steps


@pytest.fixture(scope="module")
def folder(tmp_path_factory):
    g = load_golden("qwen_asr_tiny")
    cfg, ck = qwen_setup(g)
    d = tmp_path_factory.mktemp("qwen_folder")
    meta = {"audio_pcm_scale": "32768", "max_seq_len": str(cfg.max_seq_len), "sample_rate": "16000", "special_token_ids": json.dumps(SPECIAL),
            "supported_languages": json.dumps(LANGS)}
    sub("ort_shim_qwen").export_qwen_asr_folder(str(d), cfg, ck, meta, precision=F32)
    return str(d), cfg, g


def test_graph_io_contract(folder):
    d, cfg, g = folder
    ort, wq = sub("ort_shim"), sub("ort_shim_qwen")
    L = cfg.n_layers
    s = ort.InferenceSession(f"{d}/Qwen3_ASR_Decode_Penalty_Greedy.onnx")
    ins, outs = [v.name for v in s.get_inputs()], [v.name for v in s.get_outputs()]
    assert ins[:2 * L] == [f"past_key_{i}" for i in range(L)] + [f"past_value_{i}" for i in range(L)] and not ins[2 * L].startswith("past_")
    assert outs[:2 * L] == [f"present_key_{i}" for i in range(L)] + [f"present_value_{i}" for i in range(L)] and len(outs) == 2 * L + 3
    for n in ("hidden_states", "decode_kv_seq_len", "penalty_save_id_in", "penalty_greedy_save_id_in", "penalty_penalty_value", "penalty_penalty_range"):
        assert n in ins, n
    p = ort.InferenceSession(f"{d}/Qwen3_ASR_Prefill_Greedy.onnx")
    assert [v.name for v in p.get_inputs()][2 * L:] == ["audio", "query_embed", "language_tail_embed", "prefill_history_len"] and len(p.get_outputs()) == 2 * L + 2
    k = p.get_inputs()[0]
    assert k.shape[1:4] == [cfg.n_kv_heads, 1, cfg.d_head] and isinstance(k.shape[4], str) and p.get_inputs()[L].shape[4] == cfg.d_head
    e = ort.InferenceSession(f"{d}/{wq.EMBED_FILE}.onnx")
    out = e.run(None, {"input_ids": np.array([[5, 7, 9]], np.int32)})[0]
    assert out.shape == (1, 3, cfg.d_model) and wq.embedding_as_ids(out, "t") == [5, 7, 9]
    with pytest.raises(ValueError, match="Embed graph"):
        wq.embedding_as_ids(np.zeros((1, 2, cfg.d_model), np.float32), "query_embed")


def test_reference_host_loop_matches_goldens(folder):
    d, cfg, g = folder
    cases = [c for _, c in golden_cases(g)]
    n_new = int(g["n_new"])
    host = HostLoop(d, "greedy")
    assert json.loads(host.meta["special_token_ids"])["stop"] == SPECIAL["stop"]
    for c in cases:
        audio = unit_audio(c["audio_seed"], c["n_samples"])
        # max_seq_len chosen so that generation_limit = n_new (the reference's limit rule, :673)
        toks, ids_len, steps = host.transcribe(audio, c["query_ids"].tolist(), c["language_tail_ids"].tolist(), int(c["ids_len"]) + 10 + n_new, stop=set())
        assert ids_len == int(c["ids_len"])
        if (c["margin"] > 2 * TOL).all():
            assert toks == c["token_ids"].tolist() and steps == n_new - 1
    value, rng = float(g["penalty"][0]), int(g["penalty"][1])
    host = HostLoop(d, "penalty_greedy", repeat_penalty=value, penalty_range=rng)
    for c in cases:
        if "penalty_token_ids" not in c:
            continue
        n = c["penalty_token_ids"].size
        toks, _, _ = host.transcribe(unit_audio(c["audio_seed"], c["n_samples"]), c["query_ids"].tolist(), c["language_tail_ids"].tolist(),
                                     int(c["ids_len"]) + 10 + n, stop=set())
        if (c["penalty_margin"] > 2 * TOL).all():
            assert toks == c["penalty_token_ids"].tolist()
    # stop ids end the loop and are not returned; the sampling graphs run and stay in the vocabulary
    c = cases[0]
    audio = unit_audio(c["audio_seed"], c["n_samples"])
    full = c["token_ids"].tolist()
    host = HostLoop(d, "greedy")
    cut, _, _ = host.transcribe(audio, c["query_ids"].tolist(), c["language_tail_ids"].tolist(), int(c["ids_len"]) + 10 + n_new, stop={full[2]})
    if (c["margin"] > 2 * TOL).all():
        assert cut == full[:full.index(full[2])]
    hs = HostLoop(d, "sampling", sampling=(0.8, 5, 0.9, 1.1))
    toks, _, _ = hs.transcribe(audio, [], [], int(c["ids_len"]) + 10 + 5, stop=set())
    assert len(toks) == 5 and all(0 <= t < cfg.vocab for t in toks)
