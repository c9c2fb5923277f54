"""`Data`, `Batch`, `DataLoader` with the collation rule the reference relies on (SURVEY.md
Appendix A): per key concatenate along dim 0, except keys containing 'index' or 'face' which are
concatenated along the last dim and offset by the cumulative node count."""
import re

import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith("_")]

    def __getitem__(self, k):
        return getattr(self, k)

    def __contains__(self, k):
        return k in self.__dict__

    @property
    def num_nodes(self):
        if getattr(self, "x", None) is not None:
            return self.x.size(0)
        ei = getattr(self, "edge_index", None)
        return int(ei.max()) + 1 if ei is not None and ei.numel() else 0

    @property
    def num_edges(self):
        ei = getattr(self, "edge_index", None)
        return 0 if ei is None else ei.size(1)

    def apply(self, fn):
        for k in self.keys:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, fn(v))
        return self

    def to(self, device, *a, **kw):
        return self.apply(lambda t: t.to(device, *a, **kw))

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def __repr__(self):
        return "{}({})".format(type(self).__name__, ", ".join(
            f"{k}={list(getattr(self, k).shape) if torch.is_tensor(getattr(self, k)) else getattr(self, k)}"
            for k in self.keys))


class Batch(Data):
    @staticmethod
    def from_data_list(data_list):
        keys = data_list[0].keys
        out = Batch()
        cum, batch_vec = 0, []
        cols = {k: [] for k in keys}
        for i, d in enumerate(data_list):
            n = d.num_nodes
            for k in keys:
                v = getattr(d, k)
                if torch.is_tensor(v) and re.search("(index|face)", k):
                    v = v + cum
                cols[k].append(v)
            batch_vec.append(torch.full((n,), i, dtype=torch.long))
            cum += n
        for k in keys:
            vs = cols[k]
            if torch.is_tensor(vs[0]):
                if vs[0].dim() == 0:
                    vs = [v.reshape(1) for v in vs]
                dim = -1 if re.search("(index|face)", k) else 0
                setattr(out, k, torch.cat(vs, dim=dim))
            else:
                setattr(out, k, vs)
        out.batch = torch.cat(batch_vec) if batch_vec else None
        out.num_graphs = len(data_list)
        return out


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        kwargs.pop("collate_fn", None)
        super().__init__(dataset, batch_size, shuffle,
                         collate_fn=lambda items: Batch.from_data_list(items), **kwargs)
