"""speechbrain.utils.filter_analysis mirror: window / stride bookkeeping of stacked filters
(utils/filter_analysis.py:13-237), used by the streaming feature wrapper to size its padding and caches."""
from dataclasses import dataclass


@dataclass
class FilterProperties:
    window_size: int          # input frames one output frame depends on
    stride: int = 1
    dilation: int = 1
    causal: bool = False

    def __post_init__(self):
        assert self.window_size > 0 and self.stride > 0
        assert self.dilation > 0, "Dilation must be >0. NOTE: a dilation of 1 means no dilation."

    @staticmethod
    def pointwise_filter() -> "FilterProperties":
        return FilterProperties(window_size=1, stride=1)

    def get_effective_size(self):
        return 1 + (self.window_size - 1) * self.dilation

    def get_convolution_padding(self):
        if self.window_size % 2 == 0:
            raise ValueError("Cannot determine padding with even window size")
        return self.get_effective_size() - 1 if self.causal else (self.get_effective_size() - 1) // 2

    def get_noncausal_equivalent(self):
        if not self.causal:
            return self
        return FilterProperties((self.window_size - 1) * 2 + 1, self.stride, self.dilation, False)

    def with_on_top(self, other, allow_approximate=True):
        """Properties of other(self(x)) (:143-203): sizes add through the stride, strides multiply."""
        other_size = other.window_size
        if other_size % 2 == 0:
            if not allow_approximate:
                raise ValueError("The filter to append cannot have an uneven window size. Specify "
                                 "`allow_approximate=True` if you do not need to analyze exact dependencies.")
            other_size += 1
        if self.causal != other.causal:
            if not allow_approximate:
                raise ValueError("Cannot express exact properties of causal and non-causal filters. Specify "
                                 "`allow_approximate=True` if you do not need to analyze exact dependencies.")
            return self.get_noncausal_equivalent().with_on_top(other.get_noncausal_equivalent())
        return FilterProperties(self.window_size + self.stride * (other_size - 1), self.stride * other.stride,
                                self.dilation * other.dilation, self.causal)


def stack_filter_properties(filters, allow_approximate=True):
    """[a, b, c] -> properties of c(b(a(x))); items may be modules with ``get_filter_properties()`` (:206-237)."""
    ret = FilterProperties.pointwise_filter()
    for prop in filters:
        if not isinstance(prop, FilterProperties):
            prop = prop.get_filter_properties()
        ret = ret.with_on_top(prop, allow_approximate)
    return ret
