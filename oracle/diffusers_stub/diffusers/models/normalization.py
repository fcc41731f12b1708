from .._placeholder import make_placeholder

AdaLayerNormSingle = make_placeholder("AdaLayerNormSingle")
AdaLayerNormZero = make_placeholder("AdaLayerNormZero")
AdaGroupNorm = make_placeholder("AdaGroupNorm")
