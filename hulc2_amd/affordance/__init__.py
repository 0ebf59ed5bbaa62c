"""Affordance model (BASELINE configs[4]): `hulc2.affordance.pixel_aff_lang_detector.PixelAffLangDetector` in its shipped
variant (conf/affordance/aff_detection/r3m.yaml), MI355X-native."""
from .pixel_aff_lang_detector import PixelAffLangDetector  # noqa: F401
