from .sbert_lang_encoder import SBertLang  # noqa: F401
