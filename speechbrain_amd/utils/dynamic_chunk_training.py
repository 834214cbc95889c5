"""speechbrain.utils.dynamic_chunk_training mirror: the run-time configuration object of Dynamic Chunk attention /
convolution (utils/dynamic_chunk_training.py:22-60).  The random sampler of the reference is a training-time helper
and is not part of the inference path."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class DynChunkTrainConfig:
    chunk_size: int                          # frames per chunk, > 0
    left_context_size: Optional[int] = None  # CHUNKS visible to the left (0: none, None: unlimited)

    def is_infinite_left_context(self) -> bool:
        return self.left_context_size is None

    def left_context_size_frames(self) -> Optional[int]:
        if self.left_context_size is None:
            return None
        return self.chunk_size * self.left_context_size

    # what the kernels take (include/sbk.h: chunk_size, left_chunks < 0 = unlimited)
    def kernel_args(self):
        return int(self.chunk_size), -1 if self.left_context_size is None else int(self.left_context_size)
