"""OpenAI-compatible text-chat routes over a chatbot built by `build_chatbot` — the request path of SURVEY §3.4:
FastAPI `POST /v1/chat/completions` (reference neural_chat/server/restful/textchat_api.py:482) -> `chatbot.predict` /
`chatbot.predict_stream` -> the quantised model on the GPU.

Same routes, request fields, response objects and error shape as the reference's router (textchat_api.py:374-713,
wire objects openai_protocol.py:32-230); what differs is behind them:
 * one process = one GPU = one decode engine, so requests are serialised on a lock and run on a worker thread (the
   event loop keeps answering /health); the reference's DeepSpeed fan-out to worker ports (:486-528) has no
   counterpart — tensor parallelism lives inside the engine (runtime/tp.py);
 * greedy requests (temperature 0 or top_k 1 — the OpenAI defaults of this route) ride the fused engine;
 * streamed deltas are the text pieces as generated. The reference re-splits every piece on whitespace
   (:337-345) and so drops the spaces between words; that quirk is not reproduced.
"""
import json
import threading
import time
import types
import uuid
from typing import Any, Dict, List, Optional, Union

from fastapi import APIRouter
from fastapi.concurrency import run_in_threadpool
from fastapi.responses import JSONResponse, Response, StreamingResponse
from pydantic import BaseModel as _Wire

from ...config import GenerationConfig


class ChatCompletionRequest(_Wire):
    model: str
    messages: Union[str, List[Dict[str, Any]]]
    temperature: Optional[float] = 0.7
    top_p: Optional[float] = 1.0
    top_k: Optional[int] = 1
    n: Optional[int] = 1
    max_tokens: Optional[int] = None
    stop: Optional[Union[str, List[str]]] = None
    stream: Optional[bool] = False
    repetition_penalty: Optional[float] = 1.0
    presence_penalty: Optional[float] = 0.0
    frequency_penalty: Optional[float] = 0.0
    user: Optional[str] = None


class CompletionRequest(_Wire):
    model: str
    prompt: Union[str, List[str]]
    temperature: Optional[float] = 0.7
    top_p: Optional[float] = 1.0
    top_k: Optional[int] = 1
    n: Optional[int] = 1
    max_tokens: Optional[int] = 16
    stop: Optional[Union[str, List[str]]] = None
    stream: Optional[bool] = False
    echo: Optional[bool] = False
    repetition_penalty: Optional[float] = 1.0
    presence_penalty: Optional[float] = 0.0
    frequency_penalty: Optional[float] = 0.0
    user: Optional[str] = None


def _error(status, message):
    """{"object": "error", "message", "code"} with the HTTP status as the code (textchat_api.py:104-106)."""
    return JSONResponse({"object": "error", "message": message, "code": int(status)}, status_code=int(status))


MAX_N = 8           # completions per request
MAX_TOKENS = 8192   # new tokens per completion


def _check_ranges(req):
    """Parameter ranges of textchat_api.py:56-102, same messages."""
    if req.max_tokens is not None and req.max_tokens <= 0:
        return "%s is less than the minimum of 1 - 'max_tokens'" % req.max_tokens
    if req.n is not None and req.n <= 0:
        return "%s is less than the minimum of 1 - 'n'" % req.n
    # one engine serves every request in turn: bound what a single request can make the others wait for
    if req.n is not None and req.n > MAX_N:
        return "%s is greater than the maximum of %d - 'n'" % (req.n, MAX_N)
    if req.max_tokens is not None and req.max_tokens > MAX_TOKENS:
        return "%s is greater than the maximum of %d - 'max_tokens'" % (req.max_tokens, MAX_TOKENS)
    if req.temperature is not None and req.temperature < 0:
        return "%s is less than the minimum of 0 - 'temperature'" % req.temperature
    if req.temperature is not None and req.temperature > 2:
        return "%s is greater than the maximum of 2 - 'temperature'" % req.temperature
    if req.top_p is not None and not 0 <= req.top_p <= 1:
        return "%s is outside [0, 1] - 'top_p'" % req.top_p
    if req.top_k is not None and -1 < req.top_k < 1:
        return "%s is out of Range. Either set top_k to -1 or >=1." % req.top_k
    return None


class TextChatAPIRouter(APIRouter):
    """`router.set_chatbot(chatbot)` then include the router in a FastAPI app (reference :374-390)."""

    def __init__(self):
        super().__init__()
        self.chatbot = None
        self._gpu = threading.Lock()  # one engine, one request at a time

    def set_chatbot(self, chatbot, use_deepspeed=False, world_size=1, host="0.0.0.0", port=80):
        if use_deepspeed:
            raise NotImplementedError("QBits: the DeepSpeed worker fan-out is not part of the MI355X path "
                                      "(tensor parallelism runs inside the engine)")
        self.chatbot, self.world_size, self.host, self.port = chatbot, world_size, host, port

    def get_chatbot(self):
        if self.chatbot is None:
            raise RuntimeError("Chatbot instance has not been set.")
        return self.chatbot

    # ---- request -> prompt / generation config -----------------------------------------------------------------
    def build_prompt(self, messages):
        """OpenAI messages -> the model family's prompt through its conversation template (reference
        get_generation_parameters :125-199): system -> system message, user / assistant turns in order, then an
        open assistant turn. A plain string is the prompt as it stands."""
        if isinstance(messages, str):
            return messages
        conv = self.get_chatbot().conv_template.copy()
        conv.messages = []
        for m in messages:
            role, content = m.get("role"), m.get("content")
            if isinstance(content, list):  # multimodal form: keep the text parts
                content = "\n".join(p.get("text", "") for p in content if p.get("type") == "text")
            if role == "system":
                conv.system_message = content
            elif role == "user":
                conv.append_message(conv.roles[0], content)
            elif role == "assistant":
                conv.append_message(conv.roles[1], content)
            else:
                raise ValueError("Unknown role: %s" % role)
        conv.append_message(conv.roles[1], None)
        return conv.get_prompt()

    @staticmethod
    def generation_config(req, default_max_tokens):
        greedy = not req.temperature or req.top_k == 1
        return GenerationConfig(temperature=req.temperature if req.temperature else 1.0, top_p=req.top_p,
                                top_k=req.top_k if req.top_k and req.top_k > 0 else 0,
                                repetition_penalty=req.repetition_penalty or 1.0,
                                max_new_tokens=req.max_tokens or default_max_tokens, do_sample=not greedy,
                                task="chat")

    @staticmethod
    def _stops(req, conv=None):
        stops = [req.stop] if isinstance(req.stop, str) else list(req.stop or [])
        return [s for s in stops if s]

    def _count(self, text):
        tok = getattr(self.get_chatbot(), "tokenizer", None)
        return len(tok(text).input_ids) if tok is not None and text else 0

    # ---- generation ------------------------------------------------------------------------------------------------
    def _generate(self, prompt, config, stops):
        """-> (text, finish_reason)."""
        with self._gpu:
            text = self.get_chatbot().predict(query=prompt, config=config)
        cut = min([text.find(s) for s in stops if s in text], default=-1)
        if cut >= 0:
            return text[:cut], "stop"
        return text, ("length" if self._count(text) >= config.max_new_tokens else "stop")

    def _stream(self, prompt, config, stops):
        """Text pieces until a stop string shows up (the piece is cut there). The generation itself runs in a worker
        thread that holds the engine lock only while it generates and hands pieces over a queue: a slow or vanished
        client (this generator suspended between yields, or never resumed) cannot keep the one engine locked."""
        import queue
        import threading

        # unbounded: the generation is already bounded by max_tokens, so the worker never blocks on a slow client and the
        # engine lock is held for the generation only; `done` and an exception always get through
        pieces = queue.Queue()
        gone = threading.Event()  # the consumer stopped listening (disconnect, stop string, generator closed)
        done = object()

        def work():
            try:
                with self._gpu:
                    gen, _link = self.get_chatbot().predict_stream(query=prompt, config=config)
                    if not isinstance(gen, types.GeneratorType):
                        gen = (gen,)
                    for piece in gen:
                        if gone.is_set():
                            break
                        pieces.put(piece)
                    if hasattr(gen, "close"):
                        gen.close()
                pieces.put(done)
            except BaseException as ex:  # surfaced on the consumer side
                pieces.put(ex)

        worker = threading.Thread(target=work, daemon=True)
        worker.start()
        seen = ""
        try:
            while True:
                try:
                    piece = pieces.get(timeout=1.0)
                except queue.Empty:
                    if worker.is_alive():
                        continue
                    try:  # the worker is gone: whatever it left is already in the queue
                        piece = pieces.get_nowait()
                    except queue.Empty:
                        return
                if piece is done:
                    return
                if isinstance(piece, BaseException):
                    raise piece
                if not isinstance(piece, str) or not piece:
                    continue
                seen += piece
                cut = min([seen.find(s) for s in stops if s in seen], default=-1)
                if cut >= 0:
                    keep = piece[:max(0, len(piece) - (len(seen) - cut))]
                    if keep:
                        yield keep
                    return
                yield piece
        finally:
            gone.set()


router = TextChatAPIRouter()


def _models_payload():
    name = router.get_chatbot().model_name
    now = int(time.time())
    return {"object": "list", "data": [{"id": name, "object": "model", "created": now, "owned_by": "neuralchat",
                                        "root": name, "parent": None, "permission": []}]}


@router.post("/v1/models")
async def show_available_models():
    """One model per server, as in the reference (:464-475, which answers POST)."""
    return _models_payload()


@router.get("/v1/models")
async def list_models():
    return _models_payload()


@router.get("/health")
async def health() -> Response:
    return Response(status_code=200)


def _validate(req):
    if req.model not in router.get_chatbot().model_name:  # substring match, textchat_api.py:108-115
        return _error(404, "The model `%s` does not exist." % req.model)
    bad = _check_ranges(req)
    return _error(400, bad) if bad else None


def _sse(obj):
    return "data: %s\n\n" % json.dumps(obj, ensure_ascii=False)


@router.post("/v1/chat/completions")
async def create_chat_completion(request: ChatCompletionRequest):
    """https://platform.openai.com/docs/api-reference/chat/create, as far as the reference implements it (:482-600)."""
    err = _validate(request)
    if err is not None:
        return err
    try:
        prompt = router.build_prompt(request.messages)
    except ValueError as e:
        return _error(400, str(e))
    config = router.generation_config(request, 512)
    stops = router._stops(request)
    rid, created = "chatcmpl-" + uuid.uuid4().hex[:22], int(time.time())
    if request.stream:
        def events():
            head = {"id": rid, "object": "chat.completion.chunk", "created": created, "model": request.model}
            for i in range(request.n or 1):
                yield _sse(dict(head, choices=[{"index": i, "delta": {"role": "assistant"}, "finish_reason": None}]))
                try:
                    for piece in router._stream(prompt, config, stops):
                        yield _sse(dict(head, choices=[{"index": i, "delta": {"content": piece},
                                                        "finish_reason": None}]))
                except Exception as e:  # the reference's in-band error frame (:241-244)
                    yield _sse({"text": str(e), "error_code": 500})
                    yield "data: [DONE]\n\n"
                    return
                yield _sse(dict(head, choices=[{"index": i, "delta": {}, "finish_reason": "stop"}]))
            yield "data: [DONE]\n\n"

        return StreamingResponse(events(), media_type="text/event-stream")
    choices, n_out = [], 0
    try:
        for i in range(request.n or 1):
            text, why = await run_in_threadpool(router._generate, prompt, config, stops)
            choices.append({"index": i, "message": {"role": "assistant", "content": text}, "finish_reason": why})
            n_out += router._count(text)
    except Exception as e:
        return _error(500, str(e))
    n_in = router._count(prompt) * (request.n or 1)
    return {"id": rid, "object": "chat.completion", "created": created, "model": request.model, "choices": choices,
            "usage": {"prompt_tokens": n_in, "total_tokens": n_in + n_out, "completion_tokens": n_out}}


@router.post("/v1/completions")
async def create_completion(request: CompletionRequest):
    """Plain completions (:603-713): every prompt x n, no chat template."""
    err = _validate(request)
    if err is not None:
        return err
    prompts = [request.prompt] if isinstance(request.prompt, str) else list(request.prompt)
    config = router.generation_config(request, 16)
    stops = router._stops(request)
    rid, created = "cmpl-" + uuid.uuid4().hex[:22], int(time.time())
    if request.stream:
        def events():
            head = {"id": rid, "object": "text_completion", "created": created, "model": request.model}
            idx = 0
            for p in prompts:
                for _ in range(request.n or 1):
                    if request.echo:
                        yield _sse(dict(head, choices=[{"index": idx, "text": p, "logprobs": None,
                                                        "finish_reason": None}]))
                    for piece in router._stream(p, config, stops):
                        yield _sse(dict(head, choices=[{"index": idx, "text": piece, "logprobs": None,
                                                        "finish_reason": None}]))
                    yield _sse(dict(head, choices=[{"index": idx, "text": "", "logprobs": None,
                                                    "finish_reason": "stop"}]))
                    idx += 1
            yield "data: [DONE]\n\n"

        return StreamingResponse(events(), media_type="text/event-stream")
    choices, n_in, n_out = [], 0, 0
    try:
        for p in prompts:
            for _ in range(request.n or 1):
                text, why = await run_in_threadpool(router._generate, p, config, stops)
                choices.append({"index": len(choices), "text": (p + text) if request.echo else text, "logprobs": None,
                                "finish_reason": why})
                n_in += router._count(p)
                n_out += router._count(text)
    except Exception as e:
        return _error(500, str(e))
    return {"id": rid, "object": "text_completion", "created": created, "model": request.model, "choices": choices,
            "usage": {"prompt_tokens": n_in, "total_tokens": n_in + n_out, "completion_tokens": n_out}}
