"""speechbrain.nnet.embedding mirror."""
import torch


class Embedding(torch.nn.Module):
    """nnet/embedding.py:15-125 (consider_as_one_hot=False): holder of the table; the
    decoder kernels gather from ``Embedding.weight`` directly."""

    def __init__(self, num_embeddings, embedding_dim=128, consider_as_one_hot=False, blank_id=0):
        super().__init__()
        if consider_as_one_hot:
            raise NotImplementedError("one-hot embeddings are not on the ASR path")
        self.num_embeddings, self.embedding_dim, self.blank_id = num_embeddings, embedding_dim, blank_id
        self.Embedding = torch.nn.Embedding(num_embeddings, embedding_dim)
