"""Conversation templates for the model families build_chatbot dispatches on. The reference takes these from
fastchat's `Conversation` plus its own registrations (neural_chat/prompts/prompt.py:21-52 for neural-chat-v3,
models/llama_model.py:38-44 -> "llama-2", models/mistral_model.py -> "mistral"); this is a small re-statement of the
resulting prompt strings, not of that class."""

_NEURAL_CHAT_SYSTEM = (
    "### System:\n"
    "- You are a helpful assistant chatbot trained by Intel.\n"
    "- You answer questions.\n"
    "- You are excited to be able to help the user, but will refuse to do anything that could be considered harmful "
    "to the user.\n"
    "- You are more than just an information source, you are also able to write poetry, short stories, and make "
    "jokes.</s>\n")


class Conversation:
    """roles = (user tag, assistant tag); `style` picks how turns are joined."""

    def __init__(self, name, roles, style, system_message="", sep="\n", sep2="</s>"):
        self.name, self.roles, self.style = name, roles, style
        self.system_message, self.sep, self.sep2 = system_message, sep, sep2
        self.messages = []

    def copy(self):
        c = Conversation(self.name, self.roles, self.style, self.system_message, self.sep, self.sep2)
        c.messages = [list(m) for m in self.messages]
        return c

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_prompt(self):
        if self.style == "no_colon_two":  # neural-chat v3: "<role>\n<message><sep | sep2>"
            out = self.system_message
            for i, (role, msg) in enumerate(self.messages):
                out += role + "\n" + msg + (self.sep if i % 2 == 0 else self.sep2) if msg else role + "\n"
            return out
        if self.style == "llama2":  # "[INST] <<SYS>>..<</SYS>>\n\nuser [/INST] answer </s><s>[INST] ..."
            out = ""
            for i, (role, msg) in enumerate(self.messages):
                if i % 2 == 0:
                    sys = "<<SYS>>\n%s\n<</SYS>>\n\n" % self.system_message if (i == 0 and self.system_message) else ""
                    out += "[INST] " + sys + (msg or "") + " [/INST]"
                elif msg:
                    out += " " + msg + " </s><s>"
            return out
        out = self.system_message  # "raw": plain concatenation, used for code / base models
        for role, msg in self.messages:
            out += (role + " " if role else "") + (msg or "") + (self.sep if msg else "")
        return out


_TEMPLATES = {
    "neural-chat-7b-v3": Conversation("neural-chat-7b-v3", ("### User:", "### Assistant:"), "no_colon_two",
                                      _NEURAL_CHAT_SYSTEM),
    "llama-2": Conversation("llama-2", ("[INST]", "[/INST]"), "llama2"),
    "mistral": Conversation("mistral", ("[INST]", "[/INST]"), "llama2"),
    "raw": Conversation("raw", ("", ""), "raw", sep=""),
}


def get_conv_template(name):
    return _TEMPLATES[name].copy()
