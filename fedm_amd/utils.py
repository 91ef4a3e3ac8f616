"""Small helpers, fedm/utils.py:6-35 (rank = torch.distributed rank when initialised)."""
from typing import List


def _rank():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:
        pass
    return 0


def print_rank_0(*args, **kwargs) -> None:
    if _rank() == 0:
        print(*args, **kwargs)


def comma_separated(strings: List[str]) -> str:
    return ", ".join([f"'{string}'" for string in strings])


def mesh_info(mesh) -> str:
    return (f"Number of elements is: {int(mesh.num_cells())}\n"
            f"Maximum element edge length is: {mesh.hmax():.5g}\n"
            f"Minimum element edge length is: {mesh.hmin():.5g}\n")
