"""MI355X-native hot paths of Anticipated-VINS-Mono (HIP/gfx950, FP64).

Two entry points, mirroring the reference call surfaces:
  Estimator.optimization()      -> estimator.Estimator   (vins_estimator/src/estimator.cpp:661-994)
  FeatureSelector.select()      -> feature_selector.FeatureSelector (feature_selector.cpp:74-202)
Both run through the C ABI of include/avm.h implemented by csrc/ (libavm_hip.so).
There is no CPU fallback: importing .lib without the built HIP library raises.
"""
__version__ = "0.1.0"
