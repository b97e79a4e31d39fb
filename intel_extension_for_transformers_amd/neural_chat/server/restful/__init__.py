"""REST routes of the chat server (reference: neural_chat/server/restful/): only the text-chat router is rebuilt."""
from .textchat_api import TextChatAPIRouter, router  # noqa: F401
