"""OpenAI-compatible speech endpoint on top of `Chat` (the last open piece of the host back end): the behaviour of the
reference's `examples/api/openai_api.py` -- `POST /v1/audio/speech` with the OpenAI TTS request fields, `GET /health`, a lock around the
model, WAV streaming with an open-ended RIFF header -- re-stated for this engine (reference lines cited per function; nothing is imported
from it).  What differs, deliberately:
  * the 16-bit conversion (`float_to_int16`, tools/audio/np.py:7-11, called by tools/audio/pcm.py:8-33,92-93) runs ON THE DEVICE
    (`Chat.infer(..., pcm16=True)`, csrc/codec.hip pcm16_k): the waveform crosses PCIe as int16, the bytes are the reference's bit for bit;
  * generation runs in a worker thread (starlette's thread pool), so the event loop keeps serving `/health` and other requests queue on the
    lock instead of on a blocked loop; the lock is held until a streamed response has been fully produced (the reference releases it before
    the generator is consumed, openai_api.py:211-224,265-276);
  * `response_format`: "wav" and "pcm" (raw little-endian PCM16, the OpenAI API's own name for it) always; "mp3" / "ogg" go through PyAV in
    the reference (tools/audio/av.py) and are offered only where `av` imports -- elsewhere they are refused with a 400 that says so.

    from chattts_amd.core import Chat
    from chattts_amd.server import create_app
    chat = Chat(); chat.load(custom_path=..., dtype="bf16")
    app = create_app(chat, voices={"default": spk_emb_string})        # uvicorn.run(app, ...)
"""
from __future__ import annotations

import asyncio
import io
import logging
import wave
from typing import Dict, Optional

import numpy as np

ALLOWED_PARAMS = {"model", "input", "voice", "response_format", "speed", "stream", "output_format"}    # openai_api.py:97-105
SAMPLE_RATE = 24000


def wav_stream_header(sample_rate: int = SAMPLE_RATE, bits_per_sample: int = 16, channels: int = 1) -> bytes:
    """openai_api.py:226-243: a RIFF/WAVE header whose two length fields are 0xFFFFFFFF (the length of a stream is not known)"""
    byte_rate = sample_rate * channels * bits_per_sample // 8
    block_align = channels * bits_per_sample // 8
    return (b"RIFF" + b"\xff\xff\xff\xff" + b"WAVEfmt " + (16).to_bytes(4, "little") + (1).to_bytes(2, "little")
            + channels.to_bytes(2, "little") + sample_rate.to_bytes(4, "little") + byte_rate.to_bytes(4, "little")
            + block_align.to_bytes(2, "little") + bits_per_sample.to_bytes(2, "little") + b"data" + b"\xff\xff\xff\xff")


def pcm16_to_wav_bytes(pcm: np.ndarray, sample_rate: int = SAMPLE_RATE) -> bytes:
    """tools/audio/pcm.py:8-33 for samples that are already int16: mono 16-bit RIFF/WAVE"""
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(sample_rate)
        wf.writeframes(np.ascontiguousarray(pcm, dtype="<i2").tobytes())
    return buf.getvalue()


def _av_encode(pcm: np.ndarray, fmt: str) -> bytes:
    """tools/audio/av.py: WAV bytes -> mp3 / ogg through PyAV (absent in the build container: only reached where `av` imports)"""
    import av
    src = av.open(io.BytesIO(pcm16_to_wav_bytes(pcm)), "r")
    out_buf = io.BytesIO()
    out = av.open(out_buf, "w", format=fmt)
    stream = out.add_stream({"mp3": "mp3", "ogg": "libvorbis"}[fmt], rate=SAMPLE_RATE)
    for frame in src.decode(audio=0):
        for p in stream.encode(frame):
            out.mux(p)
    for p in stream.encode(None):
        out.mux(p)
    out.close()
    src.close()
    return out_buf.getvalue()


def _have_av() -> bool:
    import importlib.util
    return importlib.util.find_spec("av") is not None


def create_app(chat, voices: Optional[Dict[str, str]] = None, logger: Optional[logging.Logger] = None, infer_kwargs: Optional[dict] = None):
    """FastAPI app serving `chat` (a loaded `chattts_amd.core.Chat`).  `voices`: OpenAI voice name -> `spk_emb` string
    (`Chat.sample_random_speaker()` / the reference's speaker files); an unknown voice falls back to "default" like openai_api.py:165.
    `infer_kwargs`: extra keywords for every `chat.infer` call (tests)."""
    from fastapi import FastAPI, HTTPException
    from fastapi.responses import JSONResponse, Response, StreamingResponse
    from pydantic import BaseModel, Field, ValidationError
    from starlette.concurrency import iterate_in_threadpool, run_in_threadpool

    log = logger or logging.getLogger("chattts_amd.server")
    voices = dict(voices or {})
    extra = dict(infer_kwargs or {})
    app = FastAPI()
    app.state.chat = chat
    app.state.model_lock = asyncio.Lock()            # openai_api.py:66
    formats = {"wav", "pcm"} | ({"mp3", "ogg"} if _have_av() else set())

    class SpeechRequest(BaseModel):                  # openai_api.py:108-127
        model: str = "tts-1"
        input: str = Field(..., max_length=2048)
        voice: Optional[str] = "default"
        response_format: Optional[str] = "wav" if "mp3" not in formats else "mp3"
        speed: Optional[float] = Field(1.0, ge=0.5, le=2.0)
        stream: Optional[bool] = False
        output_format: Optional[str] = None

    @app.exception_handler(Exception)
    async def on_error(request, exc):                # openai_api.py:141-149: one error shape
        log.error("error: %s", exc)
        return JSONResponse(status_code=getattr(exc, "status_code", 500), content={"error": {"message": str(exc), "type": exc.__class__.__name__}})

    def code_params(voice: Optional[str]):           # openai_api.py:185-205, field for field
        return chat.InferCodeParams(prompt="[speed_5]", top_P=0.5, top_K=10, temperature=0.1, repetition_penalty=1.1, max_new_token=2048,
                                    min_new_token=0, show_tqdm=False, ensure_non_empty=True, manual_seed=42,
                                    spk_emb=voices.get(voice, voices.get("default")), spk_smp=None, txt_smp=None, stream_batch=24,
                                    stream_speed=12000, pass_first_n_batches=2)

    def infer(req: "SpeechRequest"):                 # openai_api.py:168-183,207-222
        return chat.infer(text=[req.input], stream=bool(req.stream), lang=None, skip_refine_text=True, refine_text_only=False,
                          use_decoder=True, do_text_normalization=True, do_homophone_replacement=True,
                          params_infer_code=code_params(req.voice), pcm16=True, **extra)

    @app.post("/v1/audio/speech")
    async def speech(request_data: Dict):
        unknown = set(request_data) - ALLOWED_PARAMS                                 # openai_api.py:130-138
        if unknown:
            log.warning("ignoring unsupported parameters: %s", sorted(unknown))
        data = {k: request_data[k] for k in ALLOWED_PARAMS if k in request_data}
        data["model"] = "tts-1"
        try:
            req = SpeechRequest(**data)
        except ValidationError as e:
            raise HTTPException(422, detail=str(e))
        fmt = req.response_format
        if fmt not in formats:
            hint = " (mp3 / ogg need PyAV, which is not installed here)" if fmt in ("mp3", "ogg") else ""
            raise HTTPException(400, detail=f"Unsupported audio format: {fmt}, supported formats: {', '.join(sorted(formats))}{hint}")
        media = {"wav": "audio/wav", "pcm": "audio/pcm", "mp3": "audio/mpeg", "ogg": "audio/ogg"}[fmt]

        def encode(pcm: np.ndarray, header: bool) -> bytes:
            pcm = np.ascontiguousarray(np.asarray(pcm).reshape(-1), dtype="<i2")
            if fmt == "wav":
                return pcm16_to_wav_bytes(pcm) if header else pcm.tobytes()        # pcm.py:84-93
            if fmt == "pcm":
                return pcm.tobytes()
            return _av_encode(pcm, fmt)

        if req.stream:
            async def audio_stream():                                                 # openai_api.py:259-274
                async with app.state.model_lock:
                    try:
                        first = True
                        async for chunk in iterate_in_threadpool(infer(req)):
                            if fmt == "wav" and first:
                                yield wav_stream_header()
                            first = False
                            if np.asarray(chunk).size:
                                yield encode(chunk, header=False)
                    except Exception as e:      # the status line is gone by now: end the body, log the cause
                        log.error("speech synthesis failed mid-stream: %s", e)
            return StreamingResponse(audio_stream(), media_type=media)

        async with app.state.model_lock:
            try:
                wavs = await run_in_threadpool(infer, req)
            except Exception as e:
                raise HTTPException(500, detail=f"Speech synthesis failed: {e}")
        if len(wavs) == 0:
            raise HTTPException(500, detail="Speech synthesis failed: the engine returned no audio")
        body = encode(wavs[0], header=True)                                           # openai_api.py:277-288
        return Response(content=body, media_type=media, headers={"Content-Disposition": f"attachment; filename=output.{fmt}"})

    @app.get("/health")
    async def health():                                                               # openai_api.py:291-294
        loaded = bool(getattr(chat, "has_loaded", lambda: True)())
        return {"status": "healthy" if loaded else "loading", "model_loaded": loaded, "formats": sorted(formats)}

    return app
