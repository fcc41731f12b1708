from .._placeholder import make_placeholder

DualTransformer2DModel = make_placeholder("DualTransformer2DModel")
