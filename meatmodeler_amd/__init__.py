"""meatmodeler_amd — MI355X-native structure-from-motion hot path behind the call surface of
skyepurchase/MeatModeler's processor.py / bundleAdjuster.py / track.py.

    from meatmodeler_amd import processor, bundleAdjuster
    from meatmodeler_amd.track import Track

Importing `processor`, `bundleAdjuster`, `ops` or `pipeline` loads libmeatmodeler_hip.so and fails loudly if it has
not been built; there is no CPU fallback.  `synth`, `parallel`, `orb_pattern` and `track` are pure Python.
"""
__version__ = "0.1.0"
