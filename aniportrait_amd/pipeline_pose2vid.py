"""Short-clip `Pose2VideoPipeline` (src/pipelines/pipeline_pose2vid.py:286-468): the same path as the long
pipeline with a single window covering all `video_length` frames and no `context_*` kwargs.  The
reference computes the PoseGuider features once here (:396-399), which the long pipeline of this
package does as well."""
from .pipeline_pose2vid_long import Pose2VideoPipeline as _LongPipeline
from .pipeline_pose2vid_long import Pose2VideoPipelineOutput  # noqa: F401


class Pose2VideoPipeline(_LongPipeline):
    _long = False

    def __call__(self, ref_image, pose_images, ref_pose_image, width, height, video_length, num_inference_steps,
                 guidance_scale, num_images_per_prompt=1, eta=0.0, generator=None, output_type="tensor",
                 return_dict=True, callback=None, callback_steps=1, **kwargs):
        def windows_fn(L, steps):
            return [list(range(L))]

        return self._run(ref_image, pose_images, ref_pose_image, width, height, video_length, num_inference_steps,
                         guidance_scale, num_images_per_prompt, eta, generator, output_type, return_dict, callback,
                         callback_steps, windows_fn, **kwargs)
