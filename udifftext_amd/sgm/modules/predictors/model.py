"""OCR scorer placeholder.  ``ParseqPredictor`` (reference sgm/modules/predictors/model.py:7-57) runs AFTER the
denoising path (test.py:74-91) and needs the vendored PARSeq zoo plus a checkpoint; it is a "next" row of
SURVEY.md §8f, not part of this path.  The class exists so that configs naming it still parse."""


class ParseqPredictor:
    def __init__(self, ckpt_path=None):
        raise NotImplementedError(
            "ParseqPredictor (OCR scoring) is outside the MI355X denoising path; set ocr_enabled: False")
