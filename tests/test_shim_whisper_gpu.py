"""GPU: Whisper's merged graphs through the onnxruntime-API shim. The host side below issues the same API calls, in the same order, as
the reference's `_plan_merged_io` / `_probe_prefill` / `_prefill` / `_decode_tokens` (Whisper/Inference_Whisper_ONNX.py:323-663): bind
plans derived from the graphs' I/O NAMES, empty self-KV tensors at the prefill, cross-KV and self-KV values passed from one run's
outputs to the next run's inputs, two ping-pong decode bindings, per-step penalty value. Results are checked against the goldens
minted from the reference's own classes (f32 mode)."""
import numpy as np
import pytest

from conftest import sub
from helpers import golden_cases, load_golden
from test_oracle_whisper import unit_audio, whisper_setup

pytestmark = pytest.mark.gpu

F32 = 1
TOL = 1e-3


class HostLoop:
    def __init__(self, folder, strategy, repeat_penalty=1.0, penalty_range=20, sampling=(0.8, 10, 0.95, 1.0)):
        self.ort, self.io = sub("ort_shim"), sub("ort_io")
        wg = sub("ort_shim_whisper")
        self.strategy, self.repeat_penalty, self.penalty_range, self.sampling = strategy, repeat_penalty, penalty_range, sampling
        names = [wg.GRAPH_FILES[f"{role}_{strategy}"] + ".onnx" for role in ("probe_prefill", "prefill", "decode")]
        opts = self.ort.SessionOptions()
        opts.add_session_config_entry("optimization.enable_gelu_approximation", "1")
        self.probe, self.prefill_s, self.decode_s = (self.ort.InferenceSession(f"{folder}/{n}", sess_options=opts, providers=["MI355XExecutionProvider"])
                                                     for n in names)
        self.no_speech = self.ort.InferenceSession(f"{folder}/{wg.NO_SPEECH_FILE}.onnx")
        self.meta = self.ort.InferenceSession(f"{folder}/{wg.METADATA_FILE}.onnx").get_modelmeta().custom_metadata_map
        self.run_options = self.ort.RunOptions()
        self.run_options.add_run_config_entry("disable_synchronize_execution_providers", "0")
        self.plans = {"probe": self._plan(self.probe, False), "prefill": self._plan(self.prefill_s, False), "decode": self._plan(self.decode_s, True)}
        self.device = self.ort.OrtDevice(self.ort.OrtDevice.cuda(), self.ort.OrtDevice.default_memory(), 0)

    def _plan(self, session, is_decode):
        ins, outs = [m.name for m in session.get_inputs()], [m.name for m in session.get_outputs()]
        state_in = []
        for n in ins:
            if not n.startswith("in_de_"):
                break
            state_in.append(n)
        max_out = {"sampling": "sampling_sampled_id", "penalty_greedy": "greedy_max_logits_idx"}.get(self.strategy, "argmax_max_logits_idx")
        save_out = {"sampling": "sampling_save_id_out", "penalty_greedy": "greedy_save_id_out"}.get(self.strategy)
        save_in = {"sampling": ["sampling_previous_ids"], "penalty_greedy": (["penalty_save_id_in"] if is_decode else []) + ["greedy_save_id_in"]}.get(self.strategy, [])
        return dict(inputs=ins, outputs=outs, state_inputs=state_in, cross_inputs=[n for n in ins if n.startswith(("en_key_", "en_value_"))],
                    cross_outputs=[n for n in outs if n.startswith(("encoder_en_key_", "encoder_en_value_"))], max_output=max_out, save_output=save_out,
                    save_inputs=save_in, kv_seq_output="decode_kv_seq_len_next" if is_decode else "prefill_kv_seq_len",
                    sampling_inputs=[n for n in ins if n in ("sampling_temperature", "sampling_top_k", "sampling_top_p", "sampling_repetition_penalty")],
                    meta=self.io.metadata_by_name(session.get_inputs()))

    def _bind(self, binding, plan, name, value, keep, axes=None):
        v = self.ort.OrtValue.ortvalue_from_numpy(self.io.array_for(plan["meta"][name], value, axes=axes), "cpu", 0)
        keep.append(v)
        binding.bind_ortvalue_input(name, v)

    def _common_prefill_inputs(self, binding, plan, ids, keep):
        for name in plan["state_inputs"]:
            meta = plan["meta"][name]
            seq_axis = [a for a, d in enumerate(meta.shape) if a != 0 and self.io.is_dynamic_dim(d)][-1]
            self._bind(binding, plan, name, self.io.filled_for(meta, axes={0: 1, seq_axis: 0}), keep, axes={0: 1, seq_axis: 0})
        self._bind(binding, plan, "embed_input_ids", ids, keep, axes={0: ids.shape[0], 1: ids.shape[1]})
        self._bind(binding, plan, "prefill_ids_len", self.io.scalar_for(plan["meta"]["prefill_ids_len"], ids.shape[1]), keep, axes={0: 1})
        self._bind(binding, plan, "prefill_history_len", self.io.scalar_for(plan["meta"]["prefill_history_len"], 0), keep, axes={0: 1})
        for name in plan["save_inputs"]:
            self._bind(binding, plan, name, self.io.filled_for(plan["meta"][name], axes={0: 1, 1: 0}), keep, axes={0: 1, 1: 0})
        self._sampling(binding, plan, keep)
        for name in plan["outputs"]:
            binding._iobinding.bind_output(name, self.device)

    def _sampling(self, binding, plan, keep):
        vals = dict(zip(("sampling_temperature", "sampling_top_k", "sampling_top_p", "sampling_repetition_penalty"), self.sampling))
        for name in plan["sampling_inputs"]:
            self._bind(binding, plan, name, self.io.scalar_for(plan["meta"][name], vals[name]), keep, axes={0: 1})

    def probe_prefill(self, audio, ids):
        plan, binding, keep = self.plans["probe"], self.probe.io_binding(), []
        a = self.ort.OrtValue.ortvalue_from_numpy(np.ascontiguousarray(audio, dtype=np.float32).reshape(1, 1, -1), "cuda", 0)
        binding.bind_ortvalue_input("audio", a)
        self._common_prefill_inputs(binding, plan, ids, keep)
        self.probe.run_with_iobinding(binding, run_options=self.run_options)
        outs = dict(zip(plan["outputs"], binding.get_outputs()))
        cross = {n.replace("encoder_", "", 1): outs[n] for n in plan["cross_outputs"]}
        return outs, cross

    def prefill(self, ids, cross):
        plan, binding, keep = self.plans["prefill"], self.prefill_s.io_binding(), []
        for name in plan["cross_inputs"]:
            binding.bind_ortvalue_input(name, cross[name])
        self._common_prefill_inputs(binding, plan, ids, keep)
        self.prefill_s.run_with_iobinding(binding, run_options=self.run_options)
        return binding.get_outputs()

    def decode_tokens(self, prefill_outputs, cross, limit, stop_tokens):
        pp, dp = self.plans["prefill"], self.plans["decode"]
        n_state = len(dp["state_inputs"])
        pidx = {n: i for i, n in enumerate(pp["outputs"])}
        didx = {n: i for i, n in enumerate(dp["outputs"])}
        state, next_token, kv_seq = prefill_outputs[:n_state], prefill_outputs[pidx[pp["max_output"]]], prefill_outputs[pidx[pp["kv_seq_output"]]]
        selected = int(next_token.numpy().reshape(-1)[0])
        saved = prefill_outputs[pidx[pp["save_output"]]] if pp["save_output"] else None
        host_tokens, count = [], 0
        if selected not in stop_tokens and limit > 0:
            count = 1
            if saved is None:
                host_tokens.append(selected)
        bindings, keeps = [self.decode_s.io_binding(), self.decode_s.io_binding()], [[], []]
        for b, k in zip(bindings, keeps):
            for name in dp["cross_inputs"]:
                b.bind_ortvalue_input(name, cross[name])
            if "penalty_penalty_range" in dp["inputs"]:
                self._bind(b, dp, "penalty_penalty_range", self.io.scalar_for(dp["meta"]["penalty_penalty_range"], self.penalty_range), k, axes={0: 1})
            self._sampling(b, dp, k)
        pen_off = pen_on = None
        if "penalty_penalty_value" in dp["inputs"]:
            pen_off = self.ort.OrtValue.ortvalue_from_numpy(self.io.scalar_for(dp["meta"]["penalty_penalty_value"], 1.0))
            pen_on = self.ort.OrtValue.ortvalue_from_numpy(self.io.scalar_for(dp["meta"]["penalty_penalty_value"], self.repeat_penalty))
        steps = 0
        while count < limit and selected not in stop_tokens:
            b = bindings[steps & 1]
            b.bind_ortvalue_input("embed_input_ids", next_token)
            b.bind_ortvalue_input("decode_kv_seq_len", kv_seq)
            for name, value in zip(dp["state_inputs"], state):
                b.bind_ortvalue_input(name, value)
            for name in dp["save_inputs"]:
                b.bind_ortvalue_input(name, saved)
            if pen_on is not None:
                b.bind_ortvalue_input("penalty_penalty_value", pen_on if count >= self.penalty_range else pen_off)
            b.clear_binding_outputs()
            for name in dp["outputs"]:
                b._iobinding.bind_output(name, self.device)
            self.decode_s.run_with_iobinding(b, run_options=self.run_options)
            outs = b.get_outputs()
            state, next_token, kv_seq = outs[:n_state], outs[didx[dp["max_output"]]], outs[didx[dp["kv_seq_output"]]]
            selected = int(next_token.numpy().reshape(-1)[0])
            if dp["save_output"]:
                saved = outs[didx[dp["save_output"]]]
            if selected not in stop_tokens:
                count += 1
                if saved is None:
                    host_tokens.append(selected)
            steps += 1
        if saved is not None:
            host_tokens = []
            for t in saved.numpy()[0]:
                if int(t) in stop_tokens or len(host_tokens) >= limit:
                    break
                host_tokens.append(int(t))
        return host_tokens, steps


@pytest.fixture(scope="module")
def folder(tmp_path_factory):
    g = load_golden("whisper_tiny")
    cfg, ck, sup, beg = whisper_setup(str(g["cfg_name"]), int(g["ckpt_seed"]))
    d = tmp_path_factory.mktemp("whisper_folder")
    sub("ort_shim_whisper").export_whisper(str(d), cfg, ck, precision=F32, suppress_tokens=sup, begin_suppress_tokens=beg, gelu_tanh=False)
    return str(d), cfg, sup, g


def test_graph_io_contract(folder):
    d, cfg, sup, g = folder
    ort, wg = sub("ort_shim"), sub("ort_shim_whisper")
    L = cfg.n_dec_layers
    s = ort.InferenceSession(f"{d}/Whisper_DecodePenaltyGreedy.onnx")
    ins, outs = [m.name for m in s.get_inputs()], [m.name for m in s.get_outputs()]
    assert ins[:2 * L] == [f"in_de_key_layer_{i}" for i in range(L)] + [f"in_de_value_layer_{i}" for i in range(L)]
    assert outs[:2 * L] == [f"out_de_key_layer_{i}" for i in range(L)] + [f"out_de_value_layer_{i}" for i in range(L)]
    for n in ("en_key_layer_0", "en_value_layer_0", "embed_input_ids", "decode_kv_seq_len", "penalty_save_id_in", "penalty_penalty_value", "penalty_penalty_range",
              "greedy_save_id_in"):
        assert n in ins, n
    assert "logits" not in outs and "decode_kv_seq_len_next" in outs and "greedy_max_logits_idx" in outs and "greedy_save_id_out" in outs
    p = ort.InferenceSession(f"{d}/Whisper_ProbePrefillGreedy.onnx")
    pin, pout = [m.name for m in p.get_inputs()], [m.name for m in p.get_outputs()]
    assert "audio" in pin and not any(n.startswith("en_") for n in pin) and "logits" in pout and f"encoder_en_value_layer_{L - 1}" in pout
    kmeta = p.get_inputs()[0]
    assert kmeta.type == "tensor(float16)" and kmeta.shape[1:3] == [cfg.n_heads, cfg.d_head] and isinstance(kmeta.shape[3], str)
    assert sorted(wg.GRAPH_FILES.values()) == sorted(f"Whisper_{a}{b}" for a in ("ProbePrefill", "Prefill", "Decode") for b in ("Greedy", "PenaltyGreedy", "Sampling"))
    meta = ort.InferenceSession(f"{d}/ASR_Metadata.onnx").get_modelmeta().custom_metadata_map
    assert meta["audio_pcm_scale"] == "32768" and meta["max_seq_len"] == str(cfg.max_target_positions)


def test_reference_host_loop_greedy_and_penalty_match_goldens(folder):
    d, cfg, sup, g = folder
    cases = [c for _, c in golden_cases(g)]
    n_new = int(g["n_new"])
    host = HostLoop(d, "greedy")
    for c in cases:
        audio = unit_audio(c["audio_seed"], c["n_samples"])
        prompt = c["prompt"].reshape(1, -1).astype(np.int32)
        probe_outs, cross = host.probe_prefill(audio, np.array([[cfg.sot_id]], np.int32))
        lang_logits = probe_outs["logits"].numpy()
        assert lang_logits.shape == (1, cfg.vocab)
        p = float(host.no_speech.run(None, {"logits": lang_logits})[0].reshape(-1)[0])
        assert 0.0 <= p <= 1.0
        outs = host.prefill(prompt, cross)
        logits0 = outs[[m.name for m in host.prefill_s.get_outputs()].index("logits")].numpy()
        assert np.abs(logits0[0] - c["logits"][0]).max() < TOL
        toks, steps = host.decode_tokens(outs, cross, n_new, stop_tokens=set())
        if (c["margin"] > 2 * TOL).all():
            assert toks == c["token_ids"].tolist() and steps == n_new - 1
    value, rng = float(g["penalty_value"]), int(g["penalty_range"])
    host = HostLoop(d, "penalty_greedy", repeat_penalty=value, penalty_range=rng)
    for c in cases:
        audio = unit_audio(c["audio_seed"], c["n_samples"])
        _, cross = host.probe_prefill(audio, np.array([[cfg.sot_id]], np.int32))
        outs = host.prefill(c["prompt"].reshape(1, -1).astype(np.int32), cross)
        n = c["penalty_token_ids"].size
        toks, _ = host.decode_tokens(outs, cross, n, stop_tokens=set())
        if (c["penalty_margin"] > 2 * TOL).all():
            assert toks == c["penalty_token_ids"].tolist()


def test_stop_tokens_sampling_strategy_and_stale_handles(folder):
    d, cfg, sup, g = folder
    c = next(c for _, c in golden_cases(g))
    audio = unit_audio(c["audio_seed"], c["n_samples"])
    prompt = c["prompt"].reshape(1, -1).astype(np.int32)
    host = HostLoop(d, "greedy")
    _, cross = host.probe_prefill(audio, np.array([[cfg.sot_id]], np.int32))
    outs = host.prefill(prompt, cross)
    full, _ = host.decode_tokens(outs, cross, 6, stop_tokens=set())
    outs = host.prefill(prompt, cross)
    cut, _ = host.decode_tokens(outs, cross, 6, stop_tokens={full[2]})
    assert cut == full[:full.index(full[2])]
    # a KV value of an earlier run fed to a later state is refused
    old = host.prefill(prompt, cross)
    host.prefill(prompt, cross)
    with pytest.raises(ValueError, match="stale handle"):
        host.decode_tokens(old, cross, 3, stop_tokens=set())
    # sampling graphs: the scalar controls are graph inputs; runs are reproducible for the engine's seed, and ids stay in the vocabulary
    hs = HostLoop(d, "sampling", sampling=(0.8, 5, 0.9, 1.2))
    runs = []
    for _ in range(2):
        _, cross = hs.probe_prefill(audio, np.array([[cfg.sot_id]], np.int32))
        outs = hs.prefill(prompt, cross)
        runs.append(hs.decode_tokens(outs, cross, 6, stop_tokens=set())[0])
    assert runs[0] == runs[1] and len(runs[0]) == 6 and all(0 <= t < cfg.vocab for t in runs[0])
