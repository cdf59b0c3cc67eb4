"""Loading real checkpoints.  Offline there are no weights, so only local directories work."""
import os


def load_pretrained_pair(text_model_name, dna_model_name, cache_dir, device):
    for n in (text_model_name, dna_model_name):
        if not (isinstance(n, str) and os.path.isdir(n)):
            raise RuntimeError(
                f"bioreason_amd: '{n}' is not a local checkpoint directory and the HF Hub is unreachable here. "
                "Pass config objects (bioreason_amd.configs) for random-init models, or a local directory with "
                "config.json + *.safetensors."
            )
    raise NotImplementedError("safetensors loader: next row N2 (SURVEY §8f)")
