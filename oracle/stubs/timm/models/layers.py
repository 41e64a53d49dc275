from torch.nn.init import trunc_normal_  # noqa: F401
