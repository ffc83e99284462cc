"""vqa_model_builder_amd -- MI355X-native (gfx950) implementation of the AutoViVQA forward/backward hot path.

Public surface mirrors the reference's ``src.modeling.meta_arch`` / ``src.modeling.moe`` packages
(SURVEY.md section 8b); ``install_as_src()`` aliases them under those dotted names so the reference's own
training loop imports this implementation unchanged.
"""

__version__ = '0.1.0'
