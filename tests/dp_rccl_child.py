"""Child process of tests/test_gpu_distributed.py (not collected by pytest): ONE rank of a world-size-1 `nccl` (= RCCL) process
group on cuda:0 drives parler_tts_amd.distributed.generate_data_parallel with a real tiny model, so that every collective of
distributed.py - the int64 meta broadcast, one scatter per id tensor, the int64 all_reduce(MAX), the int32 / fp32 all_gather and
the length all_gather - executes on the RCCL backend with device tensors (they had only ever run on gloo).
Inputs are the reference fixture tiny_greedy, once as it is (the run fills max_length) and once with a logits processor that
forces EOS on every row from column 12 on (the run ends early: the gathered frame count is NOT max_length - K).  Prints RCCL-OK and exits 0 when codes / waveforms equal a plain generate()."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    from parler_tts_amd.distributed import broadcast_inputs, generate_data_parallel
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from tests.helpers import case_inputs, load_case, state_dict_for

    frames_seen = []
    name = "tiny_greedy"
    for force_eos_at in (None, 12):
        f, meta = load_case(name)
        cfg, sd = state_dict_for(name, meta, parts=("text_encoder", "decoder", "audio_encoder"))
        model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device=dev)
        desc, amask, prompt, _ = case_inputs(f)
        L, K, pad = int(f["max_length"]), cfg.decoder.num_codebooks, cfg.decoder.pad_token_id
        kw = dict(do_sample=False, max_length=L)
        if force_eos_at is not None:
            eos = cfg.decoder.eos_token_id

            def force_eos(ids, scores, n=force_eos_at, eos=eos):
                if ids.shape[-1] >= n:
                    scores = scores.clone()
                    scores[:, eos] = float("inf")
                return scores

            kw["logits_processor"] = [force_eos]
        wav_plain = model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, **kw)
        codes_plain = model.last_codes[0].clone()
        if force_eos_at is None:
            assert np.array_equal(codes_plain.cpu().numpy(), f["codes"]), "plain generate() left the reference fixture"
        src = dict(input_ids=desc.to(dev), attention_mask=amask.to(dev), prompt_input_ids=prompt.to(dev))

        def local_codes(**inp):
            model.generate(**inp, **kw)
            return model.last_codes[0]

        t = {}
        codes, frames = generate_data_parallel(local_codes, **src, src=0, device=dev, gather="codes", pad_token_id=pad, timings=t)
        assert codes.is_cuda and frames.is_cuda and codes.dtype == torch.long
        assert torch.equal(codes, codes_plain.to(torch.long)), f"{name}: codes through RCCL differ from plain generate()"
        assert frames.tolist() == [codes_plain.shape[-1]] * desc.shape[0]
        wav, samples = generate_data_parallel(lambda **inp: model.generate(**inp, **kw), **src, src=0, device=dev, gather="waveform")
        assert wav.is_cuda and torch.equal(wav, wav_plain.to(torch.float32).reshape(wav.shape)), f"{name}: waveforms differ"
        assert samples.tolist() == [wav_plain.reshape(desc.shape[0], -1).shape[1]] * desc.shape[0]
        full = broadcast_inputs(src, src=0, device=dev)
        assert all(torch.equal(full[k], v) for k, v in src.items())
        frames_seen.append((int(frames[0]), L - K))
        print(f"{name} (EOS forced at column {force_eos_at}): frames {int(frames[0])} of at most {L - K}, scatter {t['scatter_ms']:.2f} ms, gather {t['gather_ms']:.2f} ms", flush=True)
        model._engine.close()
    assert frames_seen[0][0] == frames_seen[0][1] and frames_seen[1][0] < frames_seen[1][1], "the EOS case must end before max_length"
    # an unsupported tensor raises (after the collective) and leaves the group usable
    try:
        broadcast_inputs(dict(x=torch.zeros(2, 1, 1, 1, 1, device=dev)), src=0, device=dev)
        raise SystemExit("5-D tensor was accepted")
    except ValueError:
        pass
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL-OK", flush=True)


if __name__ == "__main__":
    main()
