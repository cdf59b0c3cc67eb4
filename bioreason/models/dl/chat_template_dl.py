"""bioreason/models/dl/chat_template_dl.py -> bioreason_amd.chat_template"""
from bioreason_amd.chat_template import CHAT_TEMPLATE  # noqa: F401
