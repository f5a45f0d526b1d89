"""Checkpoint readers for the formats the reference loads (diffusers directory layout + peft/diffusers LoRA
safetensors): /root/reference/pipeline.py:83-109.  Pure host I/O (safetensors -> torch tensors)."""
import glob
import os
import re



def _load_dir(path):
    from safetensors.torch import load_file
    sd = {}
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError("no .safetensors under %s" % path)
    for f in files:
        sd.update(load_file(f))
    return sd


def load_flux_transformer_state_dict(path):
    """diffusers FluxTransformer2DModel checkpoint directory -> {key: tensor} (keys used as is)."""
    return _load_dir(path)


def load_vae(path, device):
    """diffusers AutoencoderKL checkpoint directory -> the HIP VAE (vae_hip.AutoencoderKL)."""
    from .synthetic import vae_param_shapes
    from .vae_hip import AutoencoderKL
    sd = _load_dir(path)
    want = vae_param_shapes()
    missing = [k for k in want if k not in sd]
    if missing:
        raise KeyError("VAE checkpoint is missing keys, e.g. %s" % missing[:5])
    for k, shp in want.items():
        if tuple(sd[k].shape) != tuple(shp):
            if sd[k].numel() == 1 or tuple(sd[k].reshape(shp).shape) != tuple(shp):
                raise ValueError("VAE parameter %s has shape %s, expected %s" % (k, tuple(sd[k].shape), shp))
            sd[k] = sd[k].reshape(shp)   # older checkpoints store the attention projections as 1x1 convs
    return AutoencoderKL({k: sd[k] for k in want}, device=device)


FULL_KEY = "__full__"   # lora_dict[FULL_KEY] = {"<module>.weight" / "<module>.bias": tensor}: whole-module overrides saved with the adapter


def load_lora_safetensors(path, default_alpha=None):
    """diffusers / peft LoRA file -> {module: (A [r,in], B [out,r])} with module names relative to the
    transformer.  Accepts 'transformer.<module>.lora_A.weight' / 'lora_B.weight' (diffusers save_lora_weights,
    trainer.py:480-490) and the '.lora.down/up.weight' spelling.  peft scaling alpha/r is folded into B.

    The reference's trainer also saves FULL copies of a few modules with every adapter (peft `modules_to_save`,
    trainer.py:297-304: x_embedder and the parameter-less norms; they reach the file through
    get_peft_model_state_dict, trainer.py:480-490).  Those tensors are returned under out[FULL_KEY] =
    {'x_embedder.weight': ..., 'x_embedder.bias': ...} and FluxDiT.set_lora swaps them in with the adapter.
    Any key that is neither a LoRA factor, an alpha nor such a module copy raises: nothing is dropped silently."""
    from safetensors import safe_open
    tensors, meta = {}, {}
    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
        for k in f.keys():
            tensors[k] = f.get_tensor(k)
    out, full, used = {}, {}, set()
    for k, v in tensors.items():
        m = re.match(r"^(?:transformer\.)?(.*)\.(lora_A|lora\.down)\.weight$", k)
        if not m:
            continue
        mod = m.group(1)
        up = k.replace("lora_A", "lora_B").replace("lora.down", "lora.up")
        if up not in tensors:
            raise KeyError("LoRA up weight missing for %s" % k)
        A, B = v, tensors[up]
        ak = k.replace(".lora_A.weight", ".alpha").replace(".lora.down.weight", ".alpha")
        alpha = tensors.get(ak)
        r = A.shape[0]
        scale = (float(alpha) / r) if alpha is not None else (float(default_alpha) / r if default_alpha else 1.0)
        out[mod] = (A, B * scale if scale != 1.0 else B)
        used.update((k, up))
        if alpha is not None:
            used.add(ak)
    for k, v in tensors.items():
        if k in used:
            continue
        # whole-module copies: '<module>.weight|bias', optionally still carrying peft's 'modules_to_save[.<adapter>]' infix
        m = re.match(r"^(?:transformer\.)?(.+?)(?:\.modules_to_save(?:\.[A-Za-z_]\w*)?)?\.(weight|bias)$", k)
        if m and "lora" not in k:
            full["%s.%s" % (m.group(1), m.group(2))] = v
            used.add(k)
    unknown = sorted(set(tensors) - used)
    if unknown:
        raise ValueError("unrecognised tensors in LoRA file %s (neither LoRA factors, alphas nor whole-module copies): %s%s"
                         % (path, unknown[:8], " ..." if len(unknown) > 8 else ""))
    if not out:
        raise ValueError("no LoRA tensors recognised in %s" % path)
    if full:
        out[FULL_KEY] = full
    return out
