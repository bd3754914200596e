"""`from smalltts.infer.utils import resample_hq` (reference src/smalltts/infer/utils.py:7-23): torch tensor in, torch tensor out,
Kaiser-windowed sinc (lowpass_filter_width 1024, rolloff 0.94, beta 14.7697); runs on the GPU when one is present."""
import torch


def resample_hq(x: torch.Tensor, sr: int, target: int) -> torch.Tensor:
    if sr == target:
        return x
    if torch.cuda.is_available():
        from smalltts_amd.api import get_engine
        y = get_engine().resample(x.reshape(-1, x.shape[-1]), sr, target)
        return y.reshape(*x.shape[:-1], y.shape[-1]).to(x.device)
    from smalltts_amd.audio import resample_hq as host_resample
    y = host_resample(x.detach().cpu().numpy().reshape(-1, x.shape[-1]), sr, target)
    return torch.from_numpy(y).reshape(*x.shape[:-1], y.shape[-1])
